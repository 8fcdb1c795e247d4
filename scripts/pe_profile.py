"""Per-kernel times of load + scan on the paired-end data set of scripts/cli_pe_check.py (6 Mb genome with planted repeats, diagnostic; GPU box).

    python scripts/pe_profile.py [N_PAIRS] [repeats] [repeat_len]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from faucet_amd import api, synth  # noqa: E402

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 2_500_000
repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rlen = int(sys.argv[3]) if len(sys.argv) > 3 else 400
g = synth.make_genome(6_000_000, 31, repeats=repeats, repeat_len=rlen)
r = synth.make_pairs(g, n_pairs, 100, 300, 30, 0.01, 32)
n = r.shape[0]
reads = torch.from_numpy(np.ascontiguousarray(r)).to("cuda:0")
tai, nh = api.load_filter_shape(8 * n, 2 * n)
batches = bench.device_batches(reads, 600_000)
ctx = api.Context(31, tai, nh, profile=True)
for rep in range(2):
    ctx.kernel_times_reset()
    ctx.load_begin()
    for b in batches:
        ctx.load_batch(b)
    ctx.load_end()
    sc = api.ReadScanner(ctx)
    st = sc.scanReads(batches)
print({k: st[k] for k in st})
print('probe outcomes [order-free, create, raise, untested]:', ctx.diag_walk_probe())
for name, (calls, ms) in sorted(ctx.kernel_times().items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"{name:24s} {ms:9.2f} ms {calls}")
