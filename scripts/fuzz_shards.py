"""The N-rank pipeline (faucet_amd/sharded.py::run_in_turn: every rank's backend calls, one rank after the other, both pass-1 protocols) on random
small inputs with RANDOM shard cuts -- uneven shards, shards of a single read, empty shards -- against the oracle's one sequential run (GPU box):
bloo1 and bloo2 after every shard, the junction records in creation order and the counters at the end.
    python scripts/fuzz_shards.py [first_seed] [last_seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faucet_amd import _lib as L  # noqa: E402
from faucet_amd import api, sharded, synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

lo = int(sys.argv[1]) if len(sys.argv) > 1 else 0
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda", 0)
bad = 0
for seed in range(lo, hi):
    rng = np.random.default_rng(40_000 + seed)
    if os.environ.get("FUZZ_VERBOSE"):
        print("seed", seed, flush=True)
    try:
        k = int(rng.choice([15, 21, 27, 31]))
        G = int(rng.integers(1500, 9000))
        g = synth.make_genome(G, seed, repeats=int(rng.integers(0, 5)), repeat_len=int(min(G // 5, rng.integers(2 * k, 8 * k))))
        n = int(rng.integers(50, 3000))
        r = synth.make_reads(g, n, int(rng.choice([60, 100, 150])), float(rng.choice([0.0, 0.01, 0.03])), seed + 1, n_rate=float(rng.choice([0.0, 0.002])))
        lines = [bytes(x) for x in np.ascontiguousarray(r)]
        world = int(rng.choice([2, 3, 5, 8]))
        cuts = sorted(int(x) for x in rng.integers(0, n + 1, size=world - 1))
        if rng.integers(0, 3) == 0 and world > 2:
            cuts[1] = cuts[0]                                   # an empty shard
        bounds = [0] + cuts + [n]
        tai, nh = 1 << int(rng.integers(13, 19)), int(rng.integers(1, 5))
        j = int(rng.integers(0, 3))
        # the oracle's sequential run, with the filters after every shard
        whole = api.ReadBatch.from_lines(lines)
        b1, b2 = po.Bloom(tai, nh), po.Bloom(tai, nh)
        after = []
        for a, b in zip(bounds[:-1], bounds[1:]):
            part = api.ReadBatch.from_lines(lines[a:b])
            if b > a:
                po.load_two_filters(b1, b2, part.bases, part.offsets, k)          # (goes on from the filters' state: the sequential run, shard by shard)
            after.append((b1.bits().copy(), b2.bits().copy()))
        osc = po.Scanner(k, j, 100, b2)
        osc.scan_reads(whole.bases, whole.offsets)
        ost = osc.stats()
        okeys, orecs = osc.junctions("creation")
        for protocol in ("presence", "fixup"):
            shards = []
            for a, b in zip(bounds[:-1], bounds[1:]):
                nb = int(rng.integers(1, 4))
                inner = sorted(set([a, b] + [int(x) for x in rng.integers(a, b + 1, size=nb - 1)])) if b > a else [a, b]
                shards.append([api.ReadBatch.from_lines(lines[x:y]) for x, y in zip(inner[:-1], inner[1:])] if b > a else [api.ReadBatch.from_lines([])])
            seen = []

            def after_load(rk, stats, t1, t2):
                seen.append((rk, np.array_equal(t1.cpu().numpy(), after[rk][0]), np.array_equal(t2.cpu().numpy(), after[rk][1])))

            merged = [0, 0]

            def after_scan(rk, stats, backend):      # (with FGPU_DEBUG_DELTA_CHECK=1 the library compares merged in-map planes with planes made again)
                d = backend.ctx.diag_prepared_refresh()
                merged[0] += d["batches_merged"]
                merged[1] += d["mismatching_words"]      # (round 6: also lk words in which a sparse link pass differed from the full pass run behind it)
                sl = backend.ctx.diag_sparse_link()
                globals()["total_sparse"] = globals().get("total_sparse", 0) + sl["windows_sparse"]
                globals()["total_full"] = globals().get("total_full", 0) + sl["windows_in_full"]

            lst, sst, last = sharded.run_in_turn(lambda: sharded.GpuShard(api.Context(k, tai, nh, j=j), dev), shards, protocol, after_load, after_scan)
            assert merged[1] == 0, (protocol, "merged planes differ from planes made again", merged)
            total_merged = globals().get("total_merged", 0) + merged[0]
            globals()["total_merged"] = total_merged
            keys, recs = last.ctx.junctions()
            last.close()
            assert all(x[1] and x[2] for x in seen) and len(seen) == world, (protocol, "filters after a shard", seen)
            assert np.array_equal(keys, okeys) and np.array_equal(recs["dist"], orecs["dist"]) and np.array_equal(recs["cov"], orecs["cov"]) and \
                np.array_equal(recs["linked"], orecs["linked"]), (protocol, "junction records")
            for key in ("n_junctions", "nb_jcheck_kmer", "nb_no_juncs", "nb_processed", "nb_skipped", "reads_no_errors", "unambiguous_reads", "reads_processed"):
                assert int(sst[key]) == int(ost[key]), (protocol, key, int(sst[key]), int(ost[key]))
    except Exception as e:   # noqa: BLE001
        bad += 1
        print("seed", seed, "FAILED", repr(e)[:500], flush=True)
print("done, seeds", lo, "to", hi - 1, "failures:", bad, "| batches whose planes were merged with the new keys:", globals().get("total_merged", 0),
      "| windows of prepared batches linked by the candidate plane:", globals().get("total_sparse", 0), "in full:", globals().get("total_full", 0))
sys.exit(1 if bad else 0)
