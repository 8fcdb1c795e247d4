"""Pass 1 of a LATER shard under the two multi-GPU protocols, timed on one GPU with the per-rank shapes of an N-rank run
(diagnostic; GPU box):   python scripts/shard_protocol_times.py [N]
presence protocol:  presence pass + load with the prefix as carried-in state
fix-up protocol:    load of the shard alone (first-set times through the shard) + fgpu_load_fixup against the prefix
Both must leave the same bloo2 for the shard."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from faucet_amd import _lib as L  # noqa: E402
from faucet_amd import api  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
reads_per_rank = 10_000_000
tai, nh = api.load_filter_shape(100_000_000 * N, 20_000_000 * N)
genome = bench.make_genome(20_000_000 * N, 2, dev)
bounds = bench.batch_bounds(reads_per_rank, 1_000_000, 2)
ctx = api.Context(31, tai, nh, profile=True)
# the prefix a middle rank sees: the k-mers of N/2 lower shards (presence passes, no ordering needed)
for r in range(max(N // 2, 1)):
    lower = bench.make_reads(genome, reads_per_rank, 100, 0.01, 1000 + r, dev)
    for b in bench.device_batches(lower, bounds):
        ctx.presence_batch(b)
    ctx.synchronize()
    del lower
prefix = torch.from_numpy(ctx.bloom_download(L.BLOO1)).to(dev)
mine = bench.make_reads(genome, reads_per_rank, 100, 0.01, 1000 + N // 2, dev)
batches = bench.device_batches(mine, bounds)
out = {}
for rep in range(2):
    # ---- presence protocol
    ctx.load_begin(); ctx.load_end()
    ctx.kernel_times_reset(); ctx.synchronize(); t0 = time.perf_counter()
    for b in batches:
        ctx.presence_batch(b)
    ctx.synchronize(); t1 = time.perf_counter()
    ctx.bloom_upload(L.BLOO1, prefix.cpu().numpy())          # (stands for the exchange + prefix-OR; not timed)
    ctx.synchronize(); t2 = time.perf_counter()
    ctx.load_begin(keep_carry=True)
    for b in batches:
        ctx.load_batch(b)
    st_a = ctx.load_end(); t3 = time.perf_counter()
    b2_a = ctx.bloom_download(L.BLOO2)
    out["presence"] = (1e3 * (t1 - t0), 1e3 * (t3 - t2), {k: round(v[1], 1) for k, v in ctx.kernel_times().items() if v[1] > 1})
    # ---- fix-up protocol
    ctx.kernel_times_reset(); ctx.synchronize(); t0 = time.perf_counter()
    ctx.load_begin(shard_times=True)
    for b in batches:
        ctx.load_batch(b)
    ctx.load_end(); t1 = time.perf_counter()
    st_b = ctx.load_fixup(prefix.data_ptr()); t2 = time.perf_counter()
    b2_b = ctx.bloom_download(L.BLOO2)
    out["fixup"] = (1e3 * (t1 - t0), 1e3 * (t2 - t1), {k: round(v[1], 1) for k, v in ctx.kernel_times().items() if v[1] > 1})
    same = bool(np.array_equal(b2_a, b2_b)) and st_a["to_bloo2"] == st_b["to_bloo2"]
    print(f"N={N} tai=2^{tai.bit_length() - 1} rep {rep}: presence {out['presence'][0]:.1f} + load {out['presence'][1]:.1f} ms | "
          f"local load {out['fixup'][0]:.1f} + fix-up {out['fixup'][1]:.1f} ms | same bloo2 and count: {same}", flush=True)
print(out, flush=True)
