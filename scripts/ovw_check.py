"""The optimistic walk of large clusters (k_ovw_round) against the oracle on repeat-rich reads, with its own statistics (GPU box; diagnostic).
    python scripts/ovw_check.py [rounds]      FGPU_OVW_ROUNDS for the run (0 = key-ordered walk only)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    os.environ["FGPU_OVW_ROUNDS"] = sys.argv[1]
os.environ.setdefault("FGPU_WALK_KO", "16")
os.environ.setdefault("FGPU_WALK_KO_ALWAYS", "1")
from faucet_amd import _lib as L  # noqa: E402
from faucet_amd import api, synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

g = synth.make_genome(300_000, 91, repeats=20, repeat_len=300)
r = synth.make_reads(g, 120_000, 100, 0.01, 92)
bases, offs = po.reads_from_matrix(r)
k, E, S = 31, 4_000_000, 1_000_000
tai, nh = api.load_filter_shape(E, S)
b1, b2 = po.Bloom(tai, nh), po.Bloom(tai, nh)
po.load_two_filters(b1, b2, bases, offs, k)
osc = po.Scanner(k, 1, 100, b2)
osc.scan_reads(bases, offs)
okeys, orecs = osc.junctions("creation")
ost = osc.stats()
n = len(offs) - 1
for span in (1 << 16, 1 << 20, 1 << 22, 0):
    ctx = api.Context(k, tai, nh, walk_window_span=span, profile=True)
    ctx.bloom_upload(L.BLOO2, b2.bits())
    cuts = np.linspace(0, n, 3).astype(int)
    parts = [api.ReadBatch(bases, offs[a:b + 1].copy()) for a, b in zip(cuts[:-1], cuts[1:])]
    t0 = time.perf_counter()
    sc = api.ReadScanner(ctx)
    sst = sc.scanReads(parts)
    dt = time.perf_counter() - t0
    keys, recs = sc.junctions()
    ok = np.array_equal(keys, okeys) and all(np.array_equal(recs[f], orecs[f]) for f in ("dist", "cov", "linked"))
    ok = ok and all(sst[c] == ost[c] for c in ("n_junctions", "nb_jcheck_kmer", "nb_no_juncs", "nb_processed", "nb_skipped"))
    kt = {n_: round(ms, 2) for n_, (c, ms) in ctx.kernel_times().items() if n_.startswith("walk")}
    print(f"span {span:8d}: {'EQUAL' if ok else 'DIFFERENT'} to the oracle, scan {1e3 * dt:7.1f} ms, windows {sst['walk_windows']}, max cluster {sst['walk_max_cluster']}, "
          f"ko/ovw pieces {sst['walk_parallel']}, ovw {ctx.diag_ovw()}, {kt}", flush=True)
    ctx.close()
