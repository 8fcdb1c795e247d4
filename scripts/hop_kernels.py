"""What one hop of the strong-scaling chain consists of, kernel by kernel: config 4 as N shards, the first two ranks' pass 2 executed in turn on one
GPU exactly as scripts/project_strong.py does (rank 0 streams and shows its table after a quarter of its reads; rank 1 prepares on that preview,
imports rank 0's table, walks), with the walk stage's kernels bracketed one by one (FGPU_PROFILE_WALK=1; the brackets cost a few per cent).
    FGPU_PROFILE_WALK=1 python scripts/hop_kernels.py [N]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from faucet_amd import _lib as L  # noqa: E402
from faucet_amd import api, sharded  # noqa: E402
from faucet_amd import synth_det as sd  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
HOP = int(sys.argv[2]) if len(sys.argv) > 2 else 1        # which rank's hop is looked at: the table it is handed = the sequential scan of shards 0 .. HOP - 1
fx = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "fullsize.json")))["config4"]
c = fx["params"]
dev = torch.device("cuda", 0)
tai, nh = api.load_filter_shape(c["E"], c["S"])
genome = sd.make_genome(c["genome"], c["genome_seed"], dev)
per = c["reads"] // N
ctx = api.Context(c["k"], tai, nh, profile=True, walk_window_span=int(os.environ.get("FAUCET_WALK_SPAN", "0")))
b = sharded.GpuShard(ctx, dev, stream_ordered=False)


def batches_of(r):
    reads = sd.make_reads(genome, per, c["read_len"], c["err"], c["read_seed"], dev, first_row=r * per)
    return reads, bench.device_batches(reads, bench.batch_bounds(per, 2_500_000, 2))


def timed(name, fn):
    ctx.kernel_times_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    ctx.synchronize()
    torch.cuda.synchronize()
    dt = 1e3 * (time.perf_counter() - t0)
    kt = sorted(ctx.kernel_times().items(), key=lambda kv: -kv[1][1])[:12]
    print(f"{name}: wall {dt:.1f} ms | " + ", ".join(f"{n} {cnt}x {ms:.1f}" for n, (cnt, ms) in kt), flush=True)
    return out


# pass 1 of the two shards the way the library does it (fix-up protocol where it can), so that bloo2 and the kept planes are a rank's own
# (FULL_BLOO2=1: pass 1 of ALL N shards, so that bloo2 -- and with it the number of junctions, the size of the table and the walk's work -- is the
# whole run's, as in scripts/project_strong.py; the default, two shards, is quicker and gives a table a third of the size)
prefix = None
b2 = None
running = None
prefixes = {}
for r in range(N if os.environ.get("FULL_BLOO2") == "1" else 2):
    reads, batches = batches_of(r)
    b.clear_filters()
    b.load(batches, keep_carry=False, shard_times=True)
    if r == 0:
        running = b.bloom_tensor(L.BLOO1).clone()
        b2 = b.bloom_tensor(L.BLOO2).clone()
    else:
        prefixes[r] = running.clone() if r == HOP else None
        mine1 = b.bloom_tensor(L.BLOO1).clone()
        b.load_fixup(running)
        b2 |= b.bloom_tensor(L.BLOO2)
        running |= mine1
        del mine1
    del reads, batches
prefix = prefixes.get(HOP)
if prefix is None:      # (HOP beyond the shards loaded: the OR of what was loaded -- only the kept planes of the hop's own load differ)
    prefix = running
hint = [None]
reads0, batches0 = batches_of(0)
b.clear_filters()
b.load(batches0, keep_carry=False, shard_times=True)
b.bloom_tensor(L.BLOO2).copy_(b2)
ctx.synchronize()
b.scan_begin()
done, marks = 0, []
for x in batches0:
    done += x.n_reads
    marks.append(done >= sharded.HINT_AFTER * per)
hi = marks.index(True)


def show(i):
    if hint[0] is None and i >= hi:
        n, buf = b.export_table(tag="hint")
        hint[0] = (buf[:max(n, 1) * L.TABLE_ENTRY_BYTES].clone(), n)


older = None
stats0 = timed("rank 0: streaming scan of its shard", lambda: b.scan_stream(batches0, after_batch=show) if HOP == 1 else None)
if HOP > 1:       # the table a later rank is handed: shards 0 .. HOP - 1 scanned one after the other by this one context (the same map, the same counters)
    for x in batches0:
        ctx.scan_batch(x)
        show(len(batches0))
    for r in range(1, HOP):
        if r == HOP - 1:      # the table rank HOP - 1 is handed: what it passes on to rank HOP as a fresher preview (round 5)
            n_o, buf_o = b.export_table(tag="older")
            older = (buf_o[:max(n_o, 1) * L.TABLE_ENTRY_BYTES].clone(), n_o)
        del batches0, reads0
        reads0, batches0 = batches_of(r)
        for x in batches0:
            ctx.scan_batch(x)
    stats0 = ctx.scan_end()
n0, buf0 = b.export_table()
table0 = buf0[:max(n0, 1) * L.TABLE_ENTRY_BYTES].clone()
del reads0, batches0
reads1, batches1 = batches_of(HOP)
b.clear_filters()
b.load(batches1, keep_carry=False, shard_times=True)
b.load_fixup(prefix)
b.bloom_tensor(L.BLOO2).copy_(b2)
ctx.synchronize()
for rep in range(int(os.environ.get("REPS", "2"))):
    b.scan_begin()
    b.import_hint(hint[0][0], hint[0][1])
    timed(f"rank {HOP}: pure stage on the preview", lambda: [b.scan_prepare(x) for x in batches1])
    if HOP > 1 and sharded.LATE_HINT and older is not None:
        timed(f"rank {HOP}: fresher preview ({older[1]} records) imported, planes of the prepared batches made again (off the chain)",
              lambda: (b.import_hint(older[0], older[1]), b.refresh_prepared()))
    carried = {n: int(stats0[n]) for n in sharded._STAT_NAMES}
    timed(f"rank {HOP}: import of {n0} records + walk of the prepared shard (the hop)", lambda: b.walk_shard(batches1, table0, n0, carried))
    timed(f"rank {HOP}: export", lambda: b.export_table())
