"""More seeds of tests/test_gpu_parity.py::test_fuzz_small_inputs_against_the_oracle than the suite runs (diagnostic; GPU box):

    python scripts/fuzz_more.py [first_seed] [last_seed]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tests.test_gpu_parity as T  # noqa: E402

lo = int(sys.argv[1]) if len(sys.argv) > 1 else 24
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 224
bad = 0
for seed in range(lo, hi):
    try:
        T.test_fuzz_small_inputs_against_the_oracle(seed)
    except Exception as e:   # noqa: BLE001
        bad += 1
        print("seed", seed, "FAILED", repr(e)[:300], flush=True)
print("done, failures:", bad)
sys.exit(1 if bad else 0)
