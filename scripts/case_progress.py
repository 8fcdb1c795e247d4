"""One data set through load + lazy scan + eager scan with a progress line after every stage (diagnostic; GPU box).
    python scripts/case_progress.py <reads> <genome> [seed] [log2 junction capacity]"""
import hashlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from faucet_amd import api  # noqa: E402

n, G = int(sys.argv[1]), int(sys.argv[2])
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = torch.device("cuda", 0)
t0 = time.perf_counter()


def say(msg):
    print(f"[{time.perf_counter() - t0:7.2f} s] {msg}", flush=True)


tai, nh = api.load_filter_shape(10 * n, 2 * n)
reads = bench.make_reads(bench.make_genome(G, 100 + seed, dev), n, 100, 0.01, 5000 + seed, dev)
batches = bench.device_batches(reads, 1_000_000)
cap = (1 << int(sys.argv[4])) if len(sys.argv) > 4 else 0
ctx = api.Context(31, tai, nh, profile=True, junction_capacity=cap)
say("data ready")
ctx.load_begin()
for i, b in enumerate(batches):
    ctx.load_batch(b)
    ctx.synchronize()
    say(f"load batch {i}")
st = ctx.load_end()
say(f"load done {st}")
ctx.scan_begin()
for i, b in enumerate(batches):
    ctx.scan_batch(b)
    ctx.synchronize()
    say(f"scan batch {i}: table entries {ctx.table_entries()}")
sst = ctx.scan_end()
say(f"scan done: junctions {sst['n_junctions']} windows {sst['walk_windows']} max cluster {sst['walk_max_cluster']} filled {sst['flags_filled']}")
say(str({k: round(v[1], 1) for k, v in ctx.kernel_times().items() if v[1] > 5}))
keys, recs = ctx.junctions()
say("junction records in creation order: sha256 " + hashlib.sha256(keys.tobytes() + recs.tobytes()).hexdigest()[:16])
