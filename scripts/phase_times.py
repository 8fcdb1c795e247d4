"""Wall time of the phases of one bench step (host clock, one synchronisation per phase boundary): where the step's time
goes besides the kernels.    gpurun -- python scripts/phase_times.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from faucet_amd import _lib as L  # noqa: E402
from faucet_amd import api  # noqa: E402

dev = torch.device("cuda", 0)
reads = bench.make_reads(bench.make_genome(20_000_000, 2, dev), 10_000_000, 100, 0.01, 1000, dev)
tai, nh = api.load_filter_shape(100_000_000, 20_000_000)
batches = bench.device_batches(reads, bench.batch_bounds(10_000_000, 1_000_000, 2))
ctx = api.Context(31, tai, nh, profile=True)
bench.step_single(ctx, batches)
for rep in range(3):
    ctx.synchronize()
    t = [time.perf_counter()]
    ctx.load_begin()
    for b in batches:
        ctx.load_batch(b)
    ctx.synchronize(); t.append(time.perf_counter())
    ctx.load_end(); t.append(time.perf_counter())
    bloo2 = ctx.bloom_download(L.BLOO2); t.append(time.perf_counter())
    ctx.scan_begin(); ctx.synchronize(); t.append(time.perf_counter())
    for b in batches:
        ctx.scan_batch(b)
    ctx.synchronize(); t.append(time.perf_counter())
    ctx.scan_end(); t.append(time.perf_counter())
    keys, recs = ctx.junctions(); t.append(time.perf_counter())
    names = ["load batches", "load_end", "bloo2 download", "scan_begin", "scan batches", "scan_end", "junction download"]
    print("  ".join(f"{n} {1e3 * (b - a):.2f}" for n, a, b in zip(names, t[:-1], t[1:])), f"| total {1e3 * (t[-1] - t[0]):.2f} ms")
