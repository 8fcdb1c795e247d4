"""BASELINE config 2 at FULL size against the oracle, bit for bit (diagnostic; GPU box; the oracle needs ~5 minutes on one core):
bloo1/bloo2 bytes, junction records in creation order, every counter.

    python scripts/full_oracle_check.py [n_reads [genome_seed read_seed]]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from faucet_amd import _lib as L  # noqa: E402
from faucet_amd import api  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
gseed = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rseed = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
dev = torch.device("cuda", 0)
reads = bench.make_reads(bench.make_genome(2 * n, gseed, dev), n, 100, 0.01, rseed, dev)
tai, nh = api.load_filter_shape(10 * n, 2 * n)
ctx = api.Context(31, tai, nh)
lst, sst, b2, keys, recs = bench.step_single(ctx, bench.device_batches(reads, 1_000_000))
b1 = ctx.bloom_download(L.BLOO1)
print("device done:", sst["n_junctions"], "junctions; running the oracle on one host core ...", flush=True)
bases, offs = po.reads_from_matrix(reads.cpu().numpy())
t0 = time.perf_counter()
ob1, ob2 = po.Bloom(tai, nh), po.Bloom(tai, nh)
olst = po.load_two_filters(ob1, ob2, bases, offs, 31)
osc = po.Scanner(31, 1, 100, ob2)
osc.scan_reads(bases, offs)
dt = time.perf_counter() - t0
okeys, orecs = osc.junctions("creation")
ost = osc.stats()
checks = {
    "bloo1": bool(np.array_equal(b1, ob1.bits())), "bloo2": bool(np.array_equal(b2, ob2.bits())),
    "to_bloo2": lst["to_bloo2"] == olst.to_bloo2, "keys (creation order)": bool(np.array_equal(keys, okeys)),
    "dist": bool(np.array_equal(recs["dist"], orecs["dist"])), "cov": bool(np.array_equal(recs["cov"], orecs["cov"])),
    "linked": bool(np.array_equal(recs["linked"], orecs["linked"])),
    "counters": all(sst[c] == ost[c] for c in ("n_junctions", "nb_jcheck_kmer", "nb_no_juncs", "nb_processed", "nb_skipped", "reads_no_errors",
                                                "unambiguous_reads", "reads_processed")),
}
print(f"oracle took {dt:.0f} s for {n} reads; windows tested inside the walk: {sst['flags_filled']}")
print(checks, "->", "PASS" if all(checks.values()) else "FAIL", flush=True)
