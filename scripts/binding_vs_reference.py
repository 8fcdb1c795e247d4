"""The reference's own contig-graph stage on top of the GPU passes, on random inputs (GPU box): oracle/_ref/faucet_ref_gpu (the compiled reference
with integration/faucet_binding.cpp linked in) against oracle/_ref/faucet_ref (the pure reference) on the runs tests/test_oracle_vs_reference_fuzz.py
draws -- same exit status and the same bytes in EVERY file, the contig files of the reference's Stage 3 included (graph nodes are named by heap
address in .fastg files: normalised as tests/test_gpu_binding.py does).    python scripts/binding_vs_reference.py [lo] [hi]"""
import os
import pathlib
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tests.test_gpu_vs_reference_fuzz as T  # noqa: E402

lo = int(sys.argv[1]) if len(sys.argv) > 1 else 7000
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 7040
bad = crashed = 0
for seed in range(lo, hi):
    with tempfile.TemporaryDirectory() as td:
        notes, rc = T.binding_differences(seed, pathlib.Path(td))
        crashed += 1 if rc < 0 else 0
        if notes:
            bad += 1
            print("seed", seed, "DIFFERENT", notes, flush=True)
print("done, seeds", lo, "to", hi - 1, "failures:", bad, "| runs in which the reference's contig graph crashed (with and without the binding alike):", crashed)
sys.exit(1 if bad else 0)
