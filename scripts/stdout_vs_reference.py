"""What the command line prints against what the compiled reference prints, line by line, up to where the reference's contig-graph stage begins
(GPU box).  Lines that carry wall-clock times are compared without their numbers.    python scripts/stdout_vs_reference.py [seed ...]"""
import difflib
import os
import pathlib
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_oracle_vs_reference_fuzz import random_run  # noqa: E402

REF, EXE = os.path.join(ROOT, "oracle", "_ref", "faucet_ref"), os.path.join(ROOT, "faucet_amd", "faucet")
seeds = [int(a) for a in sys.argv[1:]] or [100, 101, 102, 103, 104, 105]


from tests.test_gpu_vs_reference_fuzz import stdout_lines as norm  # noqa: E402


for seed in seeds:
    with tempfile.TemporaryDirectory() as td:
        tmp = pathlib.Path(td)
        path, fastq, args = random_run(seed, tmp)
        res = {}
        for tag, exe in (("ref", REF), ("gpu", EXE)):
            (tmp / tag).mkdir()
            res[tag] = subprocess.run(["stdbuf", "-o0", exe, "-read_load_file", path, "-read_scan_file", path, "-file_prefix", str(tmp / tag / "out")] + args,
                                      capture_output=True, text=True, errors="replace", timeout=600)
        a, b = norm(res["ref"].stdout), norm(res["gpu"].stdout)
        a = [ln.replace(str(tmp / "ref"), "<prefix>") for ln in a]
        b = [ln.replace(str(tmp / "gpu"), "<prefix>") for ln in b]
        cut = len(b)
        d = [ln for ln in difflib.unified_diff(a[:cut + 3], b, "reference", "command line", lineterm="", n=0) if not ln.startswith(("---", "+++", "@@"))]
        print(f"seed {seed} ({' '.join(args[-6:])}): reference printed {len(a)} lines, the command line {len(b)}; differing lines among the first {cut}: {len(d)}")
        for ln in d[:14]:
            print("     " + ln[:200])
