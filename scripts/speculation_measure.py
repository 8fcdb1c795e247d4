"""what would cross-rank speculation of the ordered walk have to replay?  (GPU box; one GPU; measurement.)

BASELINE config 2's reads cut into 8 file-order shards -- 6.25x coverage per shard, the per-shard coverage of config 4 on 8 GPUs.  The junction
table after every shard is kept (the sequential run's states T_0 .. T_7).  Shard r is then walked twice from a clean scan: on the TRUE table
T_{r-1} (what the chain of hand-overs gives rank r), and on the table ONE SHARD STALE, T_{r-2} (what rank r could start from if it did not
wait for rank r-1: the state rank r-1 itself started from).  scanInputRead's lists (every junction visit of every read) of the two walks are
compared read by read, and so are the sets of junction keys the shard creates.  A read whose list differs is a piece a speculating rank would
have to walk again once the true table arrives; what the numbers say about that is in DESIGN.md section 5.
    python scripts/speculation_measure.py [shards]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from faucet_amd import _lib as L  # noqa: E402
from faucet_amd import api, sharded  # noqa: E402
from faucet_amd import synth_det as sd  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
k, E, S, G, R, Lr = 31, 100_000_000, 20_000_000, 20_000_000, 10_000_000, 100
reads = sd.make_reads(sd.make_genome(G, 2, dev), R, Lr, 0.01, 1000, dev)
tai, nh = api.load_filter_shape(E, S)
ctx = api.Context(k, tai, nh, record_stops=True)
sh = sharded.GpuShard(ctx, dev, stream_ordered=False)
per = R // N
shards = [bench.device_batches(reads[r * per:(r + 1) * per], 625_000 if per >= 625_000 else per) for r in range(N)]
ctx.load_begin()
for s_ in shards:
    for b in s_:
        ctx.load_batch(b)
ctx.load_end()


def lists_of(shard_batches):
    """per read: tuple of the visit k-mers, in order"""
    out = []
    while True:
        t = ctx.take_stops()
        if t is None:
            break
        seq, st = t
        n_reads = shard_batches[seq].n_reads
        idx = np.searchsorted(st["read"], np.arange(n_reads + 1))
        out.append((st["ext"].copy(), idx))
    return out


def _signatures(ext, idx):
    """one 64-bit signature per read of its visit list (order-sensitive): reads with equal lists have equal signatures"""
    n = len(idx) - 1
    lens = np.diff(idx)
    if len(ext) == 0:
        return np.zeros(n, dtype=np.uint64)
    within = np.arange(len(ext), dtype=np.uint64) - np.repeat(idx[:-1].astype(np.uint64), lens)
    mixed = (ext ^ (within * np.uint64(0x9E3779B97F4A7C15))) * np.uint64(0xFF51AFD7ED558CCD)
    mixed ^= mixed >> np.uint64(33)
    cs = np.concatenate((np.zeros(1, dtype=np.uint64), np.cumsum(mixed, dtype=np.uint64)))
    return (cs[idx[1:]] - cs[idx[:-1]]) + lens.astype(np.uint64) * np.uint64(0xC4CEB9FE1A85EC53)


def walk(r, table):
    """shard r from a clean scan on `table` (n, device buffer) or on an empty map; returns (lists, junction keys after, stats)"""
    ctx.scan_begin()
    if table is not None and table[0]:
        ctx.import_table(table[1].data_ptr(), table[0])
    for b in shards[r]:
        ctx.scan_batch(b)
    st = ctx.scan_end()
    lists = lists_of(shards[r])
    keys, _ = ctx.junctions()
    n = ctx.table_entries()
    buf = torch.empty(max(n, 1) * L.TABLE_ENTRY_BYTES, dtype=torch.uint8, device=dev)
    got = ctx.export_table(buf.data_ptr(), buf.numel())
    return lists, np.asarray(keys).copy(), st, (got, buf)


# the sequential run: T_r = table after shards 0..r
tables, true_lists, true_keys = [], [], []
t = None
for r in range(N):
    lists, keys, st, t = walk(r, t)
    tables.append(t)
    true_lists.append(lists)
    true_keys.append(keys)
    print(f"shard {r}: table after it {t[0]} records", flush=True)

print("\nshard | reads | reads whose visit list differs when walked on the table one shard stale | keys created: true / stale / only-stale / only-true")
for r in range(2, N):
    t0 = time.perf_counter()
    lists, keys, st, _ = walk(r, tables[r - 2])
    differ = total = 0
    for (ea, ia), (eb, ib) in zip(true_lists[r], lists):
        total += len(ia) - 1
        differ += int((_signatures(ea, ia) != _signatures(eb, ib)).sum())
    prev = set(true_keys[r - 1].tolist())
    created_true = set(true_keys[r].tolist()) - prev
    prev2 = set(true_keys[r - 2].tolist())
    created_stale = set(keys.tolist()) - prev2
    new_true_since2 = set(true_keys[r].tolist()) - prev2           # what shards r-1 and r create together in the sequential run
    print(f"{r:5d} | {total:8d} | {differ:8d} = {100.0 * differ / total:6.2f} % | {len(created_true)} / {len(created_stale)} / "
          f"{len(created_stale - new_true_since2)} / {len(created_true - created_stale)}   ({time.perf_counter() - t0:.1f} s)", flush=True)
