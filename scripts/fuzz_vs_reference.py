"""More seeds of tests/test_gpu_vs_reference_fuzz.py than the suite runs (GPU box): the `faucet` command line against the compiled reference on random
runs, every output file byte for byte.    python scripts/fuzz_vs_reference.py [first_seed] [last_seed]"""
import os
import pathlib
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tests.test_gpu_vs_reference_fuzz as T  # noqa: E402

lo = int(sys.argv[1]) if len(sys.argv) > 1 else 200
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 500
bad = 0
for seed in range(lo, hi):
    if (seed - lo) % 20 == 0:
        print("at seed", seed, "failures so far:", bad, flush=True)      # (gpurun takes seven silent minutes for a hang)
    with tempfile.TemporaryDirectory() as td:
        try:
            T.test_cli_equals_the_compiled_reference_on_a_random_run(seed, pathlib.Path(td))
        except Exception as e:   # noqa: BLE001
            bad += 1
            print("seed", seed, "FAILED", repr(e)[:400], flush=True)
print("done, seeds", lo, "to", hi - 1, "failures:", bad)
sys.exit(1 if bad else 0)
