"""How often does the lazy-flag self-check fire?  (diagnostic; GPU box)

    python scripts/lazy_flag_frequency.py

Runs load + streaming scan on synthetic data of several sizes and seeds and reports the scans that had to be repeated with
eager junction tests (api.ReadScanner.fell_back_to_eager), and what the repeat cost.
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from faucet_amd import api  # noqa: E402

dev = torch.device("cuda", 0)
CASES = ((10_000_000, 20_000_000, 6), (20_000_000, 40_000_000, 4), (10_000_000, 100_000_000, 2))
if len(sys.argv) > 1:
    CASES = (tuple(int(x) for x in sys.argv[1:4]),)
for n, G, trials in CASES:
    tai, nh = api.load_filter_shape(10 * n, 2 * n)
    for trial in range(trials):
        reads = bench.make_reads(bench.make_genome(G, 100 + trial, dev), n, 100, 0.01, 5000 + trial, dev)
        batches = bench.device_batches(reads, 1_000_000)
        ctx = api.Context(31, tai, nh)
        ctx.load_begin()
        for b in batches:
            ctx.load_batch(b)
        ctx.load_end()
        sc = api.ReadScanner(ctx)
        t0 = time.perf_counter()
        st = sc.scanReads(batches)
        dt = time.perf_counter() - t0
        keys, recs = sc.junctions()
        import hashlib
        dig = hashlib.sha256(keys.tobytes() + recs.tobytes()).hexdigest()[:12]
        ctx.scan_set_eager(True)                      # the same scan with every junction test evaluated up front: must be identical
        st2 = sc.scanReads(batches)
        k2, r2 = sc.junctions()
        same = hashlib.sha256(k2.tobytes() + r2.tobytes()).hexdigest()[:12] == dig and all(st[c] == st2[c] for c in ("nb_processed", "nb_skipped", "nb_jcheck_kmer", "nb_no_juncs"))
        print(f"reads {n} genome {G} trial {trial}: fell back {sc.fell_back_to_eager}, scan {dt * 1e3:.0f} ms, junctions {st['n_junctions']}, "
              f"windows tested inside the walk {st['flags_filled']}, flag positions {st['flag_positions']}, identical to the eager scan: {same}", flush=True)
        ctx.close()
        del reads, batches
