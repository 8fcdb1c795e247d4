"""Pass 2 of the strong-scaling run with read shards of UNEQUAL size (the chain of walks is one sequence; what a later rank does before its turn --
its pure stage -- may take as long as the chain needs to reach it, so later shards may be larger and the first one, whose scan nobody can overlap,
smaller): BASELINE config 4 cut into the file-order shards given on the command line (millions of reads, they must add up to the fixture's reads),
every rank's stage executed in turn on ONE MI355X as scripts/project_strong.py does, transfers priced at LINK GB/s.  Pass 1 is not run per shard
here (its shards stay equal: project_strong.py); the global bloo2 comes from one ordered load of all reads.
    python scripts/project_pass2_shards.py 25,25,25,25,25,25,25,25   2.5,7.5,15,25,52.5,97.5"""
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

LINK = float(os.environ.get("LINK_GBPS", "50"))
WARM = int(os.environ.get("WARM", "2"))
fx = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "fullsize.json")))["config4"]
c = fx["params"]
dev = torch.device("cuda", 0)
tai, nh = api.load_filter_shape(c["E"], c["S"])
genome = sd.make_genome(c["genome"], c["genome_seed"], dev)
BATCH = 2_500_000


def reads_of(first, n):
    return sd.make_reads(genome, n, c["read_len"], c["err"], c["read_seed"], dev, first_row=first)


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return out, 1e3 * (time.perf_counter() - t0)


ctx = api.Context(c["k"], tai, nh)
b = sharded.GpuShard(ctx, dev, stream_ordered=False)
# the whole run's bloo2: one ordered load, 25 M reads at a time
b.clear_filters()
acc2 = None
for first in range(0, c["reads"], 25_000_000):
    n = min(25_000_000, c["reads"] - first)
    reads = reads_of(first, n)
    batches = bench.device_batches(reads, bench.batch_bounds(n, BATCH, 0))
    b.load(batches, keep_carry=True)
    ctx.synchronize()
    if acc2 is None:
        acc2 = b.bloom_tensor(L.BLOO2).clone()
    else:
        acc2 |= b.bloom_tensor(L.BLOO2)
    del reads, batches
print(f"bloo2 of the whole run: {int(torch.count_nonzero(acc2))} non-zero bytes", flush=True)

for spec in sys.argv[1:]:
    sizes = [int(round(float(x) * 1e6)) for x in spec.split(",")]
    assert sum(sizes) == c["reads"], (sum(sizes), c["reads"])
    N = len(sizes)
    firsts = [sum(sizes[:r]) for r in range(N)]
    table, n_table, stats, hint = None, 0, None, None
    t_scan0 = t_hint = 0.0
    rows = []
    for r in range(N):
        reads = reads_of(firsts[r], sizes[r])
        batches = bench.device_batches(reads, bench.batch_bounds(sizes[r], BATCH, 2))
        b.clear_filters()
        b.bloom_tensor(L.BLOO2).copy_(acc2)
        ctx.synchronize()
        prev_stats = stats
        for rep in range(WARM):
            b.scan_begin()
            if r == 0:
                done, marks = 0, []
                for x in batches:
                    done += x.n_reads
                    marks.append(done >= sharded.HINT_AFTER * sizes[0])
                hi = marks.index(True)
                hint = None
                t_begin = time.perf_counter()

                def show(i):
                    global hint, t_hint
                    if hint is None and i >= hi:
                        n, buf = b.export_table(tag="hint")
                        ctx.synchronize()
                        t_hint = 1e3 * (time.perf_counter() - t_begin)
                        hint = (buf[:max(n, 1) * L.TABLE_ENTRY_BYTES].clone(), n)
                stats, t_scan0 = timed(lambda: b.scan_stream(batches, after_batch=show))
            else:
                b.import_hint(hint[0], hint[1])
                _, ms_pure = timed(lambda: [b.scan_prepare(x) for x in batches])
                carried = {n: int(prev_stats[n]) for n in sharded._STAT_NAMES}
                stats, ms_walk = timed(lambda: b.walk_shard(batches, table, n_table, carried))
            res, ms_exp = timed(lambda: b.export_table())
        n_table, buf = res
        table = buf[:max(n_table, 1) * L.TABLE_ENTRY_BYTES].clone()
        rows.append((sizes[r], t_scan0 if r == 0 else ms_pure, 0.0 if r == 0 else ms_walk, ms_exp, n_table))
        del reads, batches
    # the critical path: rank r walks when the table has arrived AND its own pure stage (begun when the hint was there) is done
    t = rows[0][1] + rows[0][3]
    line = [f"rank 0: {rows[0][0] / 1e6:g} M reads, scan {rows[0][1]:.0f} ms (hint out at {t_hint:.0f}), table {rows[0][4]} records"]
    for r in range(1, N):
        send = rows[r - 1][4] * L.TABLE_ENTRY_BYTES / (LINK * 1e9) * 1e3
        arrive = t + send
        ready = t_hint + rows[r][1]
        start = max(arrive, ready)
        t = start + rows[r][2] + (rows[r][3] if r < N - 1 else 0.0)
        line.append(f"rank {r}: {rows[r][0] / 1e6:g} M reads, pure stage {rows[r][1]:.0f} ms (done at {ready:.0f}), table in at {arrive:.0f} (send {send:.0f}), "
                    f"import + walk {rows[r][2]:.0f}, export {rows[r][3]:.0f} -> {t:.0f}; table {rows[r][4]} records")
    print(f"shards {spec}: pass 2 = {t:.0f} ms (junctions {stats['n_junctions']})\n  " + "\n  ".join(line), flush=True)
ctx.close()
print(f"(links priced at {LINK:.0f} GB/s per direction)")
