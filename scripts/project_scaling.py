"""Projected N-GPU step of bench.py from per-rank stage times measured on ONE MI355X with the shapes of the N-rank run (GPU box):
    python scripts/project_scaling.py [N ...]          (default 2 4 8)
Every rank of `bench.py --gpus N` holds 10 M reads of an N x 20 Mb genome with filters sized for N x 1e8 k-mers.  Measured here, per N:
  pass 1   presence pass + load with a prefix as carry (presence protocol) and load alone + fix-up (fix-up protocol) of a MIDDLE rank
  pass 2   rank 0: the streaming scan of its shard;  ranks > 0: pure stage of their shard against the table hint, import of the handed-over
           table, ordered walk, export
Transfers are priced, not measured (one GPU): every rank talks to every other over its own xGMI link, LINK GB/s per link and direction.
The projection is the critical path: pass 1 + max(rank 0's scan, pure stage of the others) + (N-1) x (import + walk + export + send) + download."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from faucet_amd import _lib as L  # noqa: E402
from faucet_amd import api  # noqa: E402

LINK = float(os.environ.get("LINK_GBPS", "50"))          # sustained per link and direction (xGMI peak ~ 153 GB/s raw; RCCL p2p well below)
Ns = [int(a) for a in sys.argv[1:]] or [2, 4, 8]
dev = torch.device("cuda", 0)
n = 10_000_000
single_ms = None
rows = []
for N in Ns:
    tai, nh = api.load_filter_shape(100_000_000 * N, 20_000_000 * N)
    genome = bench.make_genome(20_000_000 * N, 2, dev)
    bounds = bench.batch_bounds(n, 1_000_000, 2)
    ctx = api.Context(31, tai, nh)
    # ---------------- pass 1 of a middle rank under both protocols
    for r in range(max(N // 2, 1)):
        lower = bench.make_reads(genome, n, 100, 0.01, 1000 + r, dev)
        for b in bench.device_batches(lower, bounds):
            ctx.presence_batch(b)
        ctx.synchronize()
        del lower
    prefix = torch.from_numpy(ctx.bloom_download(L.BLOO1)).to(dev)
    mine = bench.make_reads(genome, n, 100, 0.01, 1000 + N // 2, dev)
    batches = bench.device_batches(mine, bounds)
    best = {}
    for rep in range(2):
        ctx.load_begin(); ctx.load_end(); ctx.synchronize(); t0 = time.perf_counter()
        for b in batches:
            ctx.presence_batch(b)
        ctx.synchronize(); t1 = time.perf_counter()
        ctx.bloom_upload(L.BLOO1, prefix.cpu().numpy()); ctx.synchronize(); t2 = time.perf_counter()
        ctx.load_begin(keep_carry=True)
        for b in batches:
            ctx.load_batch(b)
        ctx.load_end(); t3 = time.perf_counter()
        best["presence"] = (1e3 * (t1 - t0), 1e3 * (t3 - t2))
        ctx.synchronize(); t0 = time.perf_counter()
        ctx.load_begin(shard_times=True)
        for b in batches:
            ctx.load_batch(b)
        ctx.load_end(); t1 = time.perf_counter()
        ctx.load_fixup(prefix.data_ptr()); t2 = time.perf_counter()
        best["fixup"] = (1e3 * (t1 - t0), 1e3 * (t2 - t1))
    del prefix
    # ---------------- pass 2: a filter that knows two shards; rank 0 streams, a later rank prepares / imports / walks / exports
    lower = bench.device_batches(bench.make_reads(genome, n, 100, 0.01, 1000, dev), bounds)
    ctx.load_begin()
    for b in lower + batches:
        ctx.load_batch(b)
    ctx.load_end()
    for rep in range(2):
        ctx.synchronize(); t0 = time.perf_counter()
        ctx.scan_begin()
        done, hint, n_hint = 0, None, 0
        for b in lower:
            ctx.scan_batch(b)
            done += b.n_reads
            if hint is None and done >= 0.25 * n:
                n_hint = ctx.table_entries()
                hint = torch.empty(max(n_hint, 1) * L.TABLE_ENTRY_BYTES, dtype=torch.uint8, device=dev)
                ctx.export_table(hint.data_ptr(), hint.numel())
        ctx.scan_end(); ctx.synchronize()
        rank0_scan = 1e3 * (time.perf_counter() - t0)
    n_in = ctx.table_entries()
    buf = torch.empty(max(n_in, 1) * L.TABLE_ENTRY_BYTES, dtype=torch.uint8, device=dev)
    ctx.export_table(buf.data_ptr(), buf.numel())
    for rep in range(2):
        ctx.synchronize(); t = [time.perf_counter()]
        ctx.scan_begin()
        ctx.import_hint(hint.data_ptr(), n_hint)
        for b in batches:
            ctx.scan_prepare(b)
        ctx.synchronize(); t.append(time.perf_counter())
        ctx.import_table(buf.data_ptr(), n_in); ctx.synchronize(); t.append(time.perf_counter())
        ctx.scan_walk_prepared(); st = ctx.scan_end(); t.append(time.perf_counter())
        out = torch.empty(max(ctx.table_entries(), 1) * L.TABLE_ENTRY_BYTES, dtype=torch.uint8, device=dev)
        ctx.export_table(out.data_ptr(), out.numel()); ctx.synchronize(); t.append(time.perf_counter())
    pure, imp, walk, exp = [1e3 * (b - a) for a, b in zip(t[:-1], t[1:])]
    n_out = int(st["n_junctions"])
    ctx.close()
    del genome, mine, batches, lower, hint, buf, out
    torch.cuda.empty_cache()
    # ---------------- projection
    B = tai // 8
    xfer_phase = 1e3 * (B / N) / (LINK * 1e9)                  # one slice to / from each peer, all links at once
    exchanges = 4 * xfer_phase                                 # prefix-OR: collect + hand back; OR-allreduce: reduce-scatter + all-gather
    p1 = {k: sum(v) for k, v in best.items()}
    proto = min(p1, key=p1.get)
    pass1 = p1[proto] + exchanges
    send = 1e3 * n_out * 32 / (LINK * 1e9)
    hop = imp + walk + exp + send
    pass2 = max(rank0_scan, pure) + (N - 1) * hop
    step = pass1 + pass2 + 3.0                                 # junction download on the last rank
    kmers = 700_000_000 * N
    rows.append((N, tai, proto, best, exchanges, rank0_scan, pure, imp, walk, exp, send, step, kmers / (step * 1e-3)))
    print(f"N={N} (2^{tai.bit_length() - 1}-bit filters): pass 1 presence {best['presence'][0]:.0f}+{best['presence'][1]:.0f} ms / fix-up {best['fixup'][0]:.0f}+{best['fixup'][1]:.0f} ms"
          f" -> {proto} {p1[proto]:.0f} ms + exchanges {exchanges:.1f} ms | pass 2: rank 0 scan {rank0_scan:.0f} ms, others' pure stage {pure:.0f} ms, per hop import {imp:.1f} + walk {walk:.1f}"
          f" + export {exp:.1f} + send {send:.1f} = {hop:.1f} ms x {N - 1} | step {step:.0f} ms = {kmers / (step * 1e-3):.3g} k-mers/s", flush=True)
print(f"(links priced at {LINK:.0f} GB/s per direction; one GPU's own step on config 2 is what bench.py --gpus 1 reports)")
