"""What a rank > 0 of an N-rank run spends per stage, on one GPU with the per-rank shapes (diagnostic; GPU box):
pure stage of all its batches with an empty junction map (scan_prepare), import of a table, ordered walk (scan_walk_prepared), export.
    python scripts/rank_stage_times.py [N]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from faucet_amd import _lib as L  # noqa: E402
from faucet_amd import api  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
n = 10_000_000
tai, nh = api.load_filter_shape(100_000_000 * N, 20_000_000 * N)
genome = bench.make_genome(20_000_000 * N, 2, dev)
bounds = bench.batch_bounds(n, 1_000_000, 2)
ctx = api.Context(31, tai, nh, profile=True)
lower = bench.device_batches(bench.make_reads(genome, n, 100, 0.01, 1000, dev), bounds)
mine = bench.device_batches(bench.make_reads(genome, n, 100, 0.01, 1001, dev), bounds)
# a filter that knows both shards, and the table the lower shard leaves behind
ctx.load_begin()
for b in lower + mine:
    ctx.load_batch(b)
ctx.load_end()
ctx.scan_begin()
done, n_hint, hint = 0, 0, None
for b in lower:
    ctx.scan_batch(b)
    done += b.n_reads
    if hint is None and done >= float(os.environ.get("HINT_AFTER", "0.25")) * n:        # what sharded.scan_sharded broadcasts: the table after a quarter of the first shard
        n_hint = ctx.table_entries()
        hint = torch.empty(max(n_hint, 1) * L.TABLE_ENTRY_BYTES, dtype=torch.uint8, device=dev)
        ctx.export_table(hint.data_ptr(), hint.numel())
ctx.scan_end()
n_in = ctx.table_entries()
buf = torch.empty(max(n_in, 1) * L.TABLE_ENTRY_BYTES, dtype=torch.uint8, device=dev)
ctx.export_table(buf.data_ptr(), buf.numel())
for rep in range(2):
    ctx.synchronize(); t = [time.perf_counter()]
    ctx.scan_begin()
    if rep:                                      # second round: with the hint
        ctx.import_hint(hint.data_ptr(), n_hint)
    for b in mine:
        ctx.scan_prepare(b)
    ctx.synchronize(); t.append(time.perf_counter())
    ctx.import_table(buf.data_ptr(), n_in)
    ctx.synchronize(); t.append(time.perf_counter())
    ctx.scan_walk_prepared()
    st = ctx.scan_end(); t.append(time.perf_counter())
    out = torch.empty(max(ctx.table_entries(), 1) * L.TABLE_ENTRY_BYTES, dtype=torch.uint8, device=dev)
    ctx.export_table(out.data_ptr(), out.numel())
    ctx.synchronize(); t.append(time.perf_counter())
    names = ["pure stage of 10 M reads (%s)" % (f"hint of {n_hint} records" if rep else "empty map"), f"import of {n_in} records", "ordered walk", f"export of {st['n_junctions']} records"]
    print(f"N={N}: junctions {st['n_junctions']} flag positions {st['flag_positions']} filled {st['flags_filled']} | " + "  ".join(f"{nm} {1e3 * (b - a):.1f} ms" for nm, a, b in zip(names, t[:-1], t[1:])), flush=True)
