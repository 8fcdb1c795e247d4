"""Stage 3's Bloom walking on the batched device probes, measured (GPU box):   python scripts/stage3_probe_rate.py [reads]
Loads and scans `reads` config-2 reads (64 MiB filters), then runs findNeighbor from every junction along every covered extension in
lock-step (faucet_amd/stage3.py) and reports walks, device calls, getValidJExtension probes and rates; plus the bare rate of
fgpu_probe_valid_extension on a large batch (each probe = 4 x (oldContains + jcheck) chains, utils/JunctionMap.cpp:474-490)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from faucet_amd import api, stage3  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
dev = torch.device("cuda", 0)
reads = bench.make_reads(bench.make_genome(n * 2, 2, dev), n, 100, 0.01, 1000, dev)
tai, nh = api.load_filter_shape(100_000_000, 20_000_000)
ctx = api.Context(31, tai, nh, profile=True)
lst, sst, b2, keys, recs = bench.step_single(ctx, bench.device_batches(reads, 500_000))
print(f"{n} reads: {len(keys)} junctions")
starts, idx = [], []
for i in range(5):
    m = (recs["dist"][:, i] > 0) & ((recs["cov"][:, i] > 0) if i < 4 else True)
    starts.append(keys[m]); idx.append(np.full(int(m.sum()), i))
starts, idx = np.concatenate(starts), np.concatenate(idx)
w = stage3.NeighborWalker(ctx, keys, recs, 31, 100)
t0 = time.perf_counter()
res = w.find_neighbors(starts, idx)
dt = time.perf_counter() - t0
print(f"findNeighbor in lock-step: {len(starts)} walks, {w.steps} device calls, {w.probes} getValidJExtension probes in {dt:.2f} s "
      f"-> {w.probes / dt:.3g} probes/s end to end (host bookkeeping in numpy and the PCIe copies of every call included); "
      f"{int((res['node'] == 1).sum())} reach a junction, {int((res['node'] == 0).sum())} end in a sink, {int(res['abort'].sum())} would trip a reference assert")
# the same walks, whole, on the device (fgpu_stage3_find_neighbors): one lane per walk, map look-ups in an HBM table
t0 = time.perf_counter()
ctx.stage3_set_junctions(keys, recs)
t1 = time.perf_counter()
got, probes = ctx.stage3_find_neighbors(starts, idx, 100)
t2 = time.perf_counter()
got, probes = ctx.stage3_find_neighbors(starts, idx, 100)
t3 = time.perf_counter()
same = all(np.array_equal(got[f], res[f]) for f in ("kmer", "node", "rindex", "dist", "len", "abort"))
print(f"whole walks on the device: map of {len(keys)} junctions set in {1e3 * (t1 - t0):.1f} ms; {len(starts)} walks, {probes} getValidJExtension probes in "
      f"{1e3 * (t3 - t2):.1f} ms (first call {1e3 * (t2 - t1):.1f} ms) -> {probes / (t3 - t2):.3g} probes/s, {len(starts) / (t3 - t2):.3g} findNeighbor/s, "
      f"host -> device -> host; results equal to the lock-step walker: {same}; kernel alone: {ctx.kernel_times().get('s3_find_neighbors')}")
big = np.repeat(starts, max(1, 8_000_000 // max(len(starts), 1)))[:8_000_000]
ctx.probe_valid_extension(big[:1000])
t0 = time.perf_counter()
ctx.probe_valid_extension(big)
dt = time.perf_counter() - t0
print(f"bare fgpu_probe_valid_extension: {len(big)} k-mers in {1e3 * dt:.1f} ms -> {len(big) / dt:.3g} getValidJExtension/s (host -> device -> host)")
