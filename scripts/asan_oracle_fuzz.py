"""The ORACLE's part of scripts/fuzz_shards.py (same seeds, same inputs, same calls: filters shard by shard, one scan) without a device, so that
it can run under AddressSanitizer here (test infrastructure checked as such: a heap error in the checker would abort a fuzz process as
surely as one in the product):
    g++ -std=c++11 -O1 -g -fPIC -shared -fsanitize=address -fno-omit-frame-pointer -o /tmp/asan/liboracle_asan.so oracle/faucet_oracle.cpp
    LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 FAUCET_ORACLE_LIB=/tmp/asan/liboracle_asan.so \
        python scripts/asan_oracle_fuzz.py 5000 5160"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faucet_amd import synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

lo = int(sys.argv[1]) if len(sys.argv) > 1 else 0
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for seed in range(lo, hi):
    rng = np.random.default_rng(40_000 + seed)
    k = int(rng.choice([15, 21, 27, 31]))
    G = int(rng.integers(1500, 9000))
    g = synth.make_genome(G, seed, repeats=int(rng.integers(0, 5)), repeat_len=int(min(G // 5, rng.integers(2 * k, 8 * k))))
    n = int(rng.integers(50, 3000))
    r = synth.make_reads(g, n, int(rng.choice([60, 100, 150])), float(rng.choice([0.0, 0.01, 0.03])), seed + 1, n_rate=float(rng.choice([0.0, 0.002])))
    lines = [bytes(x) for x in np.ascontiguousarray(r)]
    world = int(rng.choice([2, 3, 5, 8]))
    cuts = sorted(int(x) for x in rng.integers(0, n + 1, size=world - 1))
    if rng.integers(0, 3) == 0 and world > 2:
        cuts[1] = cuts[0]
    bounds = [0] + cuts + [n]
    tai, nh = 1 << int(rng.integers(13, 19)), int(rng.integers(1, 5))
    j = int(rng.integers(0, 3))
    whole = po.reads_from_lines(lines)
    b1, b2 = po.Bloom(tai, nh), po.Bloom(tai, nh)
    for a, b in zip(bounds[:-1], bounds[1:]):
        if b > a:
            bases, offs = po.reads_from_lines(lines[a:b])
            po.load_two_filters(b1, b2, bases, offs, k)
        b1.bits().copy(), b2.bits().copy()
    osc = po.Scanner(k, j, 100, b2)
    osc.scan_reads(*whole)
    osc.stats()
    osc.junctions("creation")
    del osc, b1, b2
print("oracle under ASan: seeds", lo, "to", hi - 1, "clean")
