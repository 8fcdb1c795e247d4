"""Streaming scan of the first READS reads of config 4 WITHOUT the context's own load before it (bloo2 copied in): junction count against the
oracle's checkpoint.  python scripts/stream_noload_check.py [reads in millions] [eager]"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from faucet_amd import _lib as L  # noqa: E402
from faucet_amd import api, sharded  # noqa: E402
from faucet_amd import synth_det as sd  # noqa: E402

M = float(sys.argv[1]) if len(sys.argv) > 1 else 50
eager = len(sys.argv) > 2 and sys.argv[2] == "eager"
fx = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "fullsize.json")))["config4"]
c = fx["params"]
dev = torch.device("cuda", 0)
tai, nh = api.load_filter_shape(c["E"], c["S"])
genome = sd.make_genome(c["genome"], c["genome_seed"], dev)
ctx = api.Context(c["k"], tai, nh, eager_flags=eager)
b = sharded.GpuShard(ctx, dev, stream_ordered=False)
b.clear_filters()
acc2 = None
for first in range(0, c["reads"], 25_000_000):
    reads = sd.make_reads(genome, 25_000_000, c["read_len"], c["err"], c["read_seed"], dev, first_row=first)
    batches = bench.device_batches(reads, bench.batch_bounds(25_000_000, 2_500_000, 0))
    b.load(batches, keep_carry=True)
    ctx.synchronize()
    acc2 = b.bloom_tensor(L.BLOO2).clone() if acc2 is None else acc2 | b.bloom_tensor(L.BLOO2)
    del reads, batches
n = int(M * 1e6)
reads = sd.make_reads(genome, n, c["read_len"], c["err"], c["read_seed"], dev, first_row=0)
batches = bench.device_batches(reads, bench.batch_bounds(n, 2_500_000, 2))
want = {x["reads"]: x["counters"] for x in fx["scan_checkpoints"]}
for mode in ("no load", "own load first"):
    b.clear_filters()
    if mode == "own load first":
        b.load(batches, keep_carry=False)
    b.bloom_tensor(L.BLOO2).copy_(acc2)
    ctx.synchronize()
    b.scan_begin()
    seen = []
    def after(i):
        pass
    stats = b.scan_stream(batches, after_batch=after)
    w = want.get(n)
    print(f"{mode}, {'eager' if eager else 'lazy'}: {n} reads -> junctions {stats['n_junctions']} (oracle {w['n_junctions'] if w else '?'}), "
          f"differing counters {[k for k in (w or {}) if int(stats[k]) != w[k]]}, replays so far {ctx.diag_scan_replays()}, late {ctx.diag_late_flags()}", flush=True)
