"""Long reads through the command line against the compiled reference (GPU box): read lengths far beyond the short-read configurations -- a few
hundred to tens of thousands of bases -- with errors and N, single-end, with and without cleaning; every file byte for byte."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from faucet_amd import synth  # noqa: E402

REF, EXE = os.path.join(ROOT, "oracle", "_ref", "faucet_ref"), os.path.join(ROOT, "faucet_amd", "faucet")
bad = 0
for seed, (rl, n, G) in enumerate([(500, 400, 20000), (2000, 150, 30000), (5000, 80, 40000), (20000, 30, 60000), (60000, 12, 90000), (1200, 300, 15000)]):
    rng = np.random.default_rng(500 + seed)
    g = synth.make_genome(G, 300 + seed, repeats=4, repeat_len=400)
    r = synth.make_reads(g, n, rl, 0.01, 700 + seed)
    lines = [bytes(x) for x in np.ascontiguousarray(r)]
    for _ in range(3):
        i = int(rng.integers(0, len(lines)))
        b = bytearray(lines[i]); b[int(rng.integers(0, len(b)))] = ord("N"); lines[i] = bytes(b)
    for clean in (False, True):
        with tempfile.TemporaryDirectory() as td:
            p = os.path.join(td, "in.fa")
            with open(p, "wb") as f:
                for i, s in enumerate(lines):
                    f.write(b">r%d\n" % i + s + b"\n")
            args = ["-size_kmer", "31", "-max_read_length", str(rl), "-estimated_kmers", str(4 * G), "-singletons", str(G)] + ([] if clean else ["--no_cleaning"])
            res = {}
            for tag, exe in (("ref", REF), ("gpu", EXE)):
                d = os.path.join(td, tag)
                os.mkdir(d)
                res[tag] = subprocess.run(["stdbuf", "-o0", exe, "-read_load_file", p, "-read_scan_file", p, "-file_prefix", os.path.join(d, "out")] + args,
                                          capture_output=True, text=True, errors="replace", timeout=900)
            ok = res["gpu"].returncode == (3 if clean else 0)
            files = sorted(os.listdir(os.path.join(td, "gpu")))
            for fn in files:
                a, b_ = os.path.join(td, "gpu", fn), os.path.join(td, "ref", fn)
                ok = ok and os.path.exists(b_) and open(a, "rb").read() == open(b_, "rb").read()
            dj = [ln for ln in res["gpu"].stdout.splitlines() if ln.startswith("Distinct junctions")]
            print(f"read length {rl}, {n} reads, cleaning {clean}: {'equal' if ok else 'DIFFERENT'} ({files}; {dj}) {'' if ok else res['gpu'].stderr[-300:]}", flush=True)
            bad += 0 if ok else 1
print("failures:", bad)
sys.exit(1 if bad else 0)
