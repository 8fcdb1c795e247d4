"""Per-kernel times (HIP events) of load + scan on synthetic data of a given coverage (diagnostic; GPU box).

    python scripts/high_coverage_profile.py N_READS GENOME_LENGTH
"""
import sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from faucet_amd import api
dev = torch.device("cuda", 0)
n, G = (int(x) for x in sys.argv[1:3])
tai, nh = api.load_filter_shape(10 * n, 2 * n)
reads = bench.make_reads(bench.make_genome(G, 100, dev), n, 100, 0.01, 5000, dev)
batches = bench.device_batches(reads, 1_000_000)
ctx = api.Context(31, tai, nh, profile=True, walk_window_span=int(os.environ.get("FAUCET_WALK_SPAN", "0")))
for rep in range(2):
    ctx.kernel_times_reset()
    ctx.load_begin()
    for b in batches: ctx.load_batch(b)
    ctx.load_end()
    sc = api.ReadScanner(ctx)
    st = sc.scanReads(batches)
print({k: st[k] for k in st})
for name, (calls, ms) in sorted(ctx.kernel_times().items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"{name:24s} {ms:9.2f} ms {calls}")
