"""Restart modes through the command line against the compiled reference on random inputs (GPU box): --just_load_bloom (only the .bloom is
written), then -bloom_file <that .bloom> for the scan (with and without --two_hash, which sizes the restarted filter), and --node_graph (a
Stage-3 switch that must not change the hot path's files).  Every file both write, byte for byte.   python scripts/restart_vs_reference.py [lo] [hi]"""
import os
import pathlib
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import tests.test_gpu_vs_reference_fuzz as T  # noqa: E402

lo = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 5040
bad = 0
for seed in range(lo, hi):
    with tempfile.TemporaryDirectory() as td:
        notes = T.restart_modes_differences(seed, pathlib.Path(td))
        if notes:
            bad += 1
            print("seed", seed, "DIFFERENT", notes, flush=True)
print("done, seeds", lo, "to", hi - 1, "failures:", bad)
sys.exit(1 if bad else 0)
