"""Mid-size runs through the command line against the compiled reference (GPU box; the reference takes tens of seconds each): shapes on which the
large-cluster walks and the long pair filter's fixed point have real work -- read pairs inside planted repeats at several hundred-fold
coverage -- and a thin-coverage one.  .bloom, .junctions (dump order) and both pair filters byte for byte; the reference's contig graph is
given 300 s and may be cut off or crash: its hot-path files are complete before it starts (the short pair filter may still sit in a buffer)."""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from faucet_amd import synth  # noqa: E402

REF, EXE = os.path.join(ROOT, "oracle", "_ref", "faucet_ref"), os.path.join(ROOT, "faucet_amd", "faucet")
SHAPES = [
    ("pairs in repeats, 600x", dict(G=60_000, repeats=12, repeat_len=500, pairs=180_000, rl=100, ins=300, err=0.01), ["--fastq", "--paired_ends"], 21),
    ("pairs in repeats, 250x, --mercy", dict(G=100_000, repeats=10, repeat_len=400, pairs=125_000, rl=100, ins=260, err=0.02), ["--fastq", "--paired_ends", "--mercy"], 25),
    ("single reads, 8x of 2 Mb", dict(G=2_000_000, repeats=0, repeat_len=0, pairs=80_000, rl=100, ins=400, err=0.01), [], 31),
]
bad = 0
for si, (name, p, flags, k) in enumerate(SHAPES):
    g = synth.make_genome(p["G"], 900 + si, repeats=p["repeats"], repeat_len=p["repeat_len"])
    r = synth.make_pairs(g, p["pairs"], p["rl"], p["ins"], 25, p["err"], 950 + si)
    fastq = "--fastq" in flags
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "in.fq" if fastq else "in.fa")
        (synth.write_fastq if fastq else synth.write_fasta)(path, r)
        E = max(4 * p["G"], 400_000)
        args = ["-size_kmer", str(k), "-max_read_length", str(p["rl"]), "-estimated_kmers", str(E), "-singletons", str(E // 5)] + flags
        out = {}
        for tag, exe in (("gpu", EXE), ("ref", REF)):
            d = os.path.join(td, tag)
            os.mkdir(d)
            t0 = time.time()
            try:
                rc = subprocess.run(["stdbuf", "-o0", exe, "-read_load_file", path, "-read_scan_file", path, "-file_prefix", os.path.join(d, "out")] + args,
                                    capture_output=True, text=True, errors="replace", timeout=300).returncode
            except subprocess.TimeoutExpired:
                rc = "cut off after 300 s"
            out[tag] = (rc, time.time() - t0)
        diff = []
        for fn in sorted(os.listdir(os.path.join(td, "gpu"))):
            a, b = os.path.join(td, "gpu", fn), os.path.join(td, "ref", fn)
            x = open(a, "rb").read()
            y = open(b, "rb").read() if os.path.exists(b) else None
            if y is None or (x != y and not (fn.endswith("short_pair_filter") and out["ref"][0] != 0 and x[:len(y)] == y)):
                diff.append((fn, len(x), None if y is None else len(y)))
        print(f"{name} ({2 * p['pairs']} reads, k = {k}): {'equal' if not diff else 'DIFFERENT ' + str(diff)} | command line {out['gpu'][1]:.1f} s (exit {out['gpu'][0]}), "
              f"reference {out['ref'][1]:.1f} s (exit {out['ref'][0]}) | files {sorted(os.listdir(os.path.join(td, 'gpu')))}", flush=True)
        bad += 1 if diff else 0
print("failures:", bad)
sys.exit(1 if bad else 0)
