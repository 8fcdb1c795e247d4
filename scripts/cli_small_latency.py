"""Wall time of the `faucet` command line on the reference's own small case (config 1: 1 000 reads), run on the GPU box.

    PYTHONPATH=. python scripts/cli_small_latency.py [repeats]

A user of the reference who tries the tool on a test file sees start-up cost, not throughput: process start, HIP initialisation,
the context's allocations, two tiny passes, the output files.  FGPU_CLI_TIMES=1 makes the CLI print its own phase clock on stderr.
"""
import gzip
import json
import os
import subprocess
import sys
import tempfile
import time

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = os.path.join(root, "tests", "golden", "c1_k21")
meta = json.load(open(os.path.join(d, "case.json")))
tmp = tempfile.mkdtemp()
reads = os.path.join(tmp, "reads.fa")
open(reads, "wb").write(gzip.open(os.path.join(d, "reads.fa.gz")).read())
args = [a if not a.endswith(".fa") else reads for a in meta["args"]]
exe = os.path.join(root, "faucet_amd", "faucet")
env = dict(os.environ, FGPU_CLI_TIMES="1")
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    t0 = time.perf_counter()
    r = subprocess.run([exe, "-read_load_file", reads, "-read_scan_file", reads, "-file_prefix", os.path.join(tmp, "out")] + args,
                       capture_output=True, text=True, env=env)
    dt = time.perf_counter() - t0
    assert r.returncode == 0, r.stderr
    print(f"run {i}: {dt * 1e3:8.1f} ms wall")
    for line in r.stderr.splitlines():
        if line.startswith(("[cli]", "[fgpu_create]")):
            print("    " + line)
