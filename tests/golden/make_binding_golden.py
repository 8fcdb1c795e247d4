#!/usr/bin/env python3
"""Digests of EVERYTHING the pure compiled reference writes on two small goldens with cleaning on -- the hot path's files and what its own
contig-graph stage makes of them -- for tests/test_gpu_binding.py, which runs the reference with integration/faucet_binding.cpp linked in
against libfaucet_gpu.so on the GPU box (where /root/reference does not exist) and must end with the same files.

    make -C oracle ref && python tests/golden/make_binding_golden.py
"""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests.golden_util import Case  # noqa: E402

PURE = os.path.join(ROOT, "oracle", "_ref", "faucet_ref")


def normalised(path):
    data = open(path, "rb").read()
    if not path.endswith(".fastg"):          # the reference names graph nodes by heap address
        return data
    seen = {}
    return re.sub(rb"0x[0-9a-f]+", lambda m: seen.setdefault(m.group(0), b"n%d" % len(seen)), data)


out = {}
for name in ("se_cleaning_k21", "pe_fastq_k21", "c1_k21"):
    c = Case(name)
    with tempfile.TemporaryDirectory() as td:
        inp = os.path.join(td, "reads.fq" if c.fastq else "reads.fa")
        with open(inp, "wb") as f:
            f.write(c.reads_text())
        r = subprocess.run([PURE, "-read_load_file", inp, "-read_scan_file", inp, "-file_prefix", os.path.join(td, "out")] + c.meta["args"],
                           capture_output=True, text=True)
        files = {f: hashlib.sha256(normalised(os.path.join(td, f))).hexdigest() for f in sorted(os.listdir(td)) if f.startswith("out.")}
        out[name] = {"exit": r.returncode, "files": files,
                     "summary": [ln for ln in r.stdout.splitlines() if ln.startswith(("Distinct junctions:", "Number of kmers that we j-checked:",
                                                                                        "Number of processed kmers:", "Number of skipped kmers:"))]}
with open(os.path.join(HERE, "binding_stage3.json"), "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
    f.write("\n")
print(json.dumps(out, indent=1)[:1500])
