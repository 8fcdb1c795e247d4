"""More seeds of tests/test_gpu_vs_reference_fuzz.py::test_cli_over_read_shards_... than the suite runs (GPU box): `faucet -gpus N` (the C++ host over N
read shards) against the compiled reference on random runs, every output file and the log.    python scripts/fuzz_shards_vs_reference.py [first_seed] [last_seed]
N cycles through 2, 3, 4, 5, 7; odd seeds force the fix-up protocol's fail planes (FAUCET_SHARD_PLANES=1), every fourth the presence protocol."""
import os
import pathlib
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tests.test_gpu_vs_reference_fuzz as T  # noqa: E402

lo = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 1100
bad = 0
for seed in range(lo, hi):
    n = (2, 3, 4, 5, 7)[seed % 5]
    os.environ.pop("FAUCET_SHARD_PLANES", None)
    os.environ.pop("FAUCET_SHARD_PROTOCOL", None)
    if seed % 4 == 0:
        os.environ["FAUCET_SHARD_PROTOCOL"] = "presence"
    elif seed % 2:
        os.environ["FAUCET_SHARD_PLANES"] = "1"
    with tempfile.TemporaryDirectory() as td:
        try:
            T.test_cli_equals_the_compiled_reference_on_a_random_run(seed, pathlib.Path(td), cli_extra=["-gpus", str(n)])
        except Exception as e:   # noqa: BLE001
            bad += 1
            print("seed", seed, "gpus", n, "FAILED", repr(e)[:600], flush=True)
    if (seed - lo) % 20 == 19:
        print("... seeds up to", seed, "failures so far:", bad, flush=True)
print("done, seeds", lo, "to", hi - 1, "failures:", bad)
sys.exit(1 if bad else 0)
