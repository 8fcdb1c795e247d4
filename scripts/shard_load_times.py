"""Where a rank's pass 1 of the strong mode goes: config 4's first shard of N (25 M reads at N = 8) through presence and load on one context,
wall time against kernel times (HIP events).  python scripts/shard_load_times.py [N]"""
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
fx = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "fullsize.json")))["config4"]
c = fx["params"]
dev = torch.device("cuda", 0)
tai, nh = api.load_filter_shape(c["E"], c["S"])
genome = sd.make_genome(c["genome"], c["genome_seed"], dev)
per = c["reads"] // N
reads = sd.make_reads(genome, per, c["read_len"], c["err"], c["read_seed"], dev, first_row=0)
batches = bench.device_batches(reads, bench.batch_bounds(per, 2_500_000, 2))
ctx = api.Context(c["k"], tai, nh, profile=True)
b = sharded.GpuShard(ctx, dev, stream_ordered=False)


def run(name, fn, reps=3):
    for i in range(reps):
        ctx.kernel_times_reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        ctx.synchronize()
        torch.cuda.synchronize()
        dt = 1e3 * (time.perf_counter() - t0)
        kt = sorted(ctx.kernel_times().items(), key=lambda kv: -kv[1][1])[:8]
        print(f"{name} run {i}: wall {dt:.1f} ms | kernels " + ", ".join(f"{n} {cnt}x {ms:.1f}" for n, (cnt, ms) in kt) + f" | sum {sum(v[1] for v in ctx.kernel_times().values()):.1f}", flush=True)


run("presence", lambda: (b.clear_filters(), [b.presence(x) for x in batches]))
run("load keep_carry=True, empty carry", lambda: (b.clear_filters(), b.load(batches, keep_carry=True)))
run("load keep_carry=False", lambda: b.load(batches, keep_carry=False))
run("load begin/end only (no batches), keep_carry=True", lambda: (b.clear_filters(), b.load([], keep_carry=True)))
run("clear_filters", lambda: b.clear_filters())
# a higher rank: shard 1 loaded on the presence bitmap of shard 0 (what rank 1 of N does)
b.clear_filters()
[b.presence(x) for x in batches]
ctx.synchronize()
pres0 = b.bloom_tensor(L.BLOO1).clone()
del batches, reads
reads = sd.make_reads(genome, per, c["read_len"], c["err"], c["read_seed"], dev, first_row=per)
batches = bench.device_batches(reads, bench.batch_bounds(per, 2_500_000, 2))
run("rank 1: load keep_carry=True on rank 0's presence", lambda: (b.clear_filters(), b.bloom_tensor(L.BLOO1).copy_(pres0), ctx.synchronize(), b.load(batches, keep_carry=True)))
# the other protocol for the same rank: own load with times that count through the shard, then the fix-up on the lower ranks' bits
ok = b.fixup_possible(batches)
print("fix-up possible:", ok, flush=True)
if ok:
    run("rank 1: own load with shard times", lambda: (b.clear_filters(), b.load(batches, keep_carry=False, shard_times=True)), reps=2)
    run("rank 1: fix-up on rank 0's bits", lambda: b.load_fixup(pres0), reps=1)
