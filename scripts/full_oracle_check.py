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
repeats = int(sys.argv[4]) if len(sys.argv) > 4 else 0      # planted copies of 500-base blocks: real branching, not only error junctions
Lr = int(sys.argv[5]) if len(sys.argv) > 5 else 100
err = float(sys.argv[6]) if len(sys.argv) > 6 else 0.01
two_hash = len(sys.argv) > 7 and sys.argv[7] == "two_hash"
dev = torch.device("cuda", 0)
if repeats:
    from faucet_amd import synth
    g = synth.make_genome(2 * n, gseed)
    rng = np.random.default_rng(gseed + 1)
    for _ in range(repeats // 4):           # families of 4 copies, each copy with a few private mutations
        src = int(rng.integers(0, 2 * n - 500))
        block = g[src:src + 500].copy()
        for _c in range(4):
            dst = int(rng.integers(0, 2 * n - 500))
            cp = block.copy()
            m = rng.random(500) < 0.01
            cp[m] = synth._ACGT[rng.integers(0, 4, size=int(m.sum()))]
            g[dst:dst + 500] = cp
    genome = torch.from_numpy(g).to(dev)
else:
    genome = bench.make_genome(2 * n, gseed, dev)
reads = bench.make_reads(genome, n, Lr, err, rseed, dev)
if two_hash:
    _, tai, nh = api.size_two_hash(40 * n, 0.04)
else:
    tai, nh = api.load_filter_shape(10 * n, 2 * n)
ctx = api.Context(31, tai, nh)
# the step bench.py times: ramped batches, outputs into page-locked buffers, bloo2 copied while the scan runs
lst, sst, b2, keys, recs = bench.step_single(ctx, bench.device_batches(reads, bench.batch_bounds(n, 1_000_000, 2)), pinned=True)
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
