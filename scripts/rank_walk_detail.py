import os, sys, time, torch
sys.path.insert(0, "/root/repo")
os.environ["FGPU_PROFILE_WALK"] = "1"
os.environ["FGPU_NO_OVERLAP"] = "1"
import bench
from faucet_amd import _lib as L, api
N = int(sys.argv[1])
dev = torch.device("cuda", 0)
n = 10_000_000
tai, nh = api.load_filter_shape(100_000_000 * N, 20_000_000 * N)
genome = bench.make_genome(20_000_000 * N, 2, dev)
bounds = bench.batch_bounds(n, 1_000_000, 2)
ctx = api.Context(31, tai, nh, profile=True)
lower = bench.device_batches(bench.make_reads(genome, n, 100, 0.01, 1000, dev), bounds)
mine = bench.device_batches(bench.make_reads(genome, n, 100, 0.01, 1001, dev), bounds)
ctx.load_begin()
for b in lower + mine: ctx.load_batch(b)
ctx.load_end()
ctx.scan_begin()
for b in lower: ctx.scan_batch(b)
ctx.scan_end()
n_in = ctx.table_entries()
buf = torch.empty(max(n_in, 1) * L.TABLE_ENTRY_BYTES, dtype=torch.uint8, device=dev)
ctx.export_table(buf.data_ptr(), buf.numel())
for rep in range(2):
    ctx.scan_begin()
    for b in mine: ctx.scan_prepare(b)
    ctx.synchronize(); ctx.kernel_times_reset()
    ctx.import_table(buf.data_ptr(), n_in); ctx.synchronize(); t0 = time.perf_counter()
    ctx.scan_walk_prepared(); st = ctx.scan_end(); t1 = time.perf_counter()
print(f"N={N}: ordered walk of a prepared shard {1e3*(t1-t0):.1f} ms, windows {st['walk_windows']}, followers {st['walk_followers']}, max cluster {st['walk_max_cluster']}")
print({k: round(v[1], 2) for k, v in sorted(ctx.kernel_times().items(), key=lambda kv: -kv[1][1]) if v[1] > 0.2})
