"""Pass 1 alone on config 2's reads, with the kernels' times (GPU box; used by scripts/first_table_experiment.sh -- the measurement builds give wrong
filters, so nothing is scanned)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from faucet_amd import api  # noqa: E402

dev = torch.device("cuda", 0)
n = 10_000_000
tai, nh = api.load_filter_shape(100_000_000, 20_000_000)
reads = bench.make_reads(bench.make_genome(20_000_000, 2, dev), n, 100, 0.01, 1000, dev)
batches = bench.device_batches(reads, bench.batch_bounds(n, 1_000_000, 2))
ctx = api.Context(31, tai, nh, profile=True)
for rep in range(4):
    ctx.kernel_times_reset()
    ctx.load_begin()
    for b in batches:
        ctx.load_batch(b)
    st = ctx.load_end()
t = ctx.kernel_times()
g = lambda k: t.get(k, (0, 0.0))[1]
print(f"{sys.argv[1] if len(sys.argv) > 1 else '':14s} load_mark {g('load_mark'):6.2f}  load_resolve {g('load_resolve'):6.2f}  carry_update {g('carry_update'):5.2f}  "
      f"sum {g('load_mark') + g('load_resolve') + g('carry_update'):6.2f} ms per pass  (to_bloo2 {st['to_bloo2']})")
