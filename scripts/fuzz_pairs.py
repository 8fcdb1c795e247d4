"""Paired-end fuzz (diagnostic; GPU box): random small genomes with repeats, ragged reads with N and empty records, random batchings and window
sizes, pair filters small enough for collisions -- both pair filters, the pair counts, the junction records and the scan's counters against the
oracle's scanReads with paired_ends (src/ReadScanner.cpp:284-359).  Run it also with the large-cluster walks forced:
    FGPU_WALK_KO=2 FGPU_WALK_KO_ALWAYS=1 python scripts/fuzz_pairs.py [first_seed] [last_seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faucet_amd import _lib as L  # noqa: E402
from faucet_amd import api, synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
import tests.test_gpu_parity as T  # noqa: E402

lo = int(sys.argv[1]) if len(sys.argv) > 1 else 0
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
for seed in range(lo, hi):
    rng = np.random.default_rng(7000 + seed)
    try:
        k = int(rng.integers(9, 32))
        G = int(rng.integers(600, 6000))
        g = synth.make_genome(G, seed, repeats=int(rng.integers(0, 7)), repeat_len=int(min(G // 5, rng.integers(2 * k, 12 * k))))
        n_pairs = int(rng.integers(1, 4000))
        rl = int(rng.choice([40, 75, 100, 150]))
        ins = int(rng.integers(rl, 3 * rl))
        r = synth.make_pairs(g, n_pairs, rl, ins, int(rng.integers(0, 30)), float(rng.choice([0.0, 0.01, 0.03])), seed + 1)
        lines = [bytes(x) for x in np.ascontiguousarray(r)]
        for _ in range(int(rng.integers(0, 6))):           # N inside reads, truncated reads, empty records (each still toggles firstEnd)
            i = int(rng.integers(0, len(lines)))
            what = int(rng.integers(0, 3))
            if what == 0:
                b = bytearray(lines[i]); b[int(rng.integers(0, len(b)))] = ord("N"); lines[i] = bytes(b)
            elif what == 1:
                lines[i] = lines[i][:int(rng.integers(0, len(lines[i])))]
            else:
                lines.insert(i, b"")
        whole = api.ReadBatch.from_lines(lines)
        bases, offs = whole.bases, whole.offsets
        E = int(rng.choice([2_000, 20_000, 200_000]))
        tai, nh = 1 << int(rng.integers(12, 20)), int(rng.integers(1, 5))
        no_cleaning = bool(rng.integers(0, 4) == 0)
        b1, b2, lst, _ = T.oracle_run((bases, offs), k, tai, nh, 1, 100)
        _, stai, snh = api.size_optimal(max(E // 20, 64), np.float32(0.01))
        _, ltai, lnh = api.size_optimal(max(E // 10, 64), np.float32(0.01))
        short, long_ = po.Bloom(stai, snh), po.Bloom(ltai, lnh)
        osc = po.Scanner(k, 1, 100, b2, short_pf=short, long_pf=long_)
        osc.scan_reads(bases, offs, paired_ends=True, no_cleaning=no_cleaning)
        ost = osc.stats()
        ctx = api.Context(k, tai, nh, record_stops=True, walk_window_span=int(rng.choice([0, 64, 1000, 1 << 14, 1 << 18])))
        ctx.bloom_upload(L.BLOO2, b2.bits())
        if no_cleaning:
            ctx.scan_long_pairs(0, 0, 1)                  # --no_cleaning: only the loop's two counts
        else:
            ctx.scan_short_pairs(short.tai, short.n_hash, False)
            ctx.scan_long_pairs(long_.tai, long_.n_hash, 2)
        cuts = sorted(set([0, len(lines)] + [int(x) for x in rng.integers(0, len(lines) + 1, size=int(rng.integers(0, 6)))]))
        ctx.scan_begin()
        for a, b in zip(cuts[:-1], cuts[1:]):
            ctx.scan_batch(api.ReadBatch.from_lines(lines[a:b]))
        sst = ctx.scan_end()
        bits, empty, not_empty = ctx.scan_long_pairs_download(0 if no_cleaning else long_.tai)
        assert (empty, not_empty) == (ost["empty_count"], ost["not_empty_count"]), ("pair counts", empty, not_empty, ost["empty_count"], ost["not_empty_count"])
        if not no_cleaning:
            assert np.array_equal(bits, long_.bits()), "long pair filter"
            assert np.array_equal(ctx.scan_short_pairs_download(short.tai), short.bits()), "short pair filter"
        T._scan_equals_oracle(ctx, sst, osc)
        ctx.close()
    except Exception as e:   # noqa: BLE001
        bad += 1
        print("seed", seed, "FAILED", repr(e)[:400], flush=True)
print("done, seeds", lo, "to", hi - 1, "failures:", bad)
sys.exit(1 if bad else 0)
