"""Shapes that stress the ordered walk's scheduling rather than its size, through the command line against the compiled reference (GPU box): very
high coverage of a tiny genome, low-complexity reads (homopolymer and dinucleotide runs: one k-mer many times on a read), a hundred thousand
copies of one read, tandem repeats.  Files byte for byte, and the command line's pass times."""
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from faucet_amd import synth  # noqa: E402

REF, EXE = os.path.join(ROOT, "oracle", "_ref", "faucet_ref"), os.path.join(ROOT, "faucet_amd", "faucet")
rng = np.random.default_rng(77)


def reads_of(g, n, rl, err, seed):
    return [bytes(x) for x in np.ascontiguousarray(synth.make_reads(g, n, rl, err, seed))]


g_small = synth.make_genome(30_000, 11, repeats=0, repeat_len=0)
g_tandem = np.concatenate([synth.make_genome(5_000, 12, repeats=0, repeat_len=0)] + [np.frombuffer(b"ACGTTGCA" * 40, dtype=np.uint8)] * 30 + [synth.make_genome(5_000, 13, repeats=0, repeat_len=0)])
low = []
for i in range(60_000):
    u = rng.integers(0, 3)
    low.append((b"A" * 100) if u == 0 else (b"AT" * 50) if u == 1 else bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=100)))
SHAPES = [
    ("5000x of a 30 kb genome, single end", reads_of(g_small, 150_000 * 10, 100, 0.01, 21)[:600_000], [], 31),
    ("low-complexity reads (A..., ATAT..., random)", low, [], 21),
    ("one read 100 000 times", [reads_of(g_small, 1, 100, 0.0, 22)[0]] * 100_000, [], 25),
    ("tandem repeats (an 8-mer 1 200 times in the genome), 200x", reads_of(g_tandem, 40_000, 100, 0.01, 23), ["--paired_ends"], 21),
]
bad = 0
for name, lines, flags, k in SHAPES:
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "in.fa")
        with open(path, "wb") as f:
            for i, s in enumerate(lines):
                f.write(b">r%d\n" % i + s + b"\n")
        args = ["-size_kmer", str(k), "-max_read_length", "100", "-estimated_kmers", "2000000", "-singletons", "400000"] + flags
        out = {}
        for tag, exe in (("gpu", EXE), ("ref", REF)):
            d = os.path.join(td, tag)
            os.mkdir(d)
            t0 = time.time()
            try:
                r = subprocess.run(["stdbuf", "-o0", exe, "-read_load_file", path, "-read_scan_file", path, "-file_prefix", os.path.join(d, "out")] + args,
                                   capture_output=True, text=True, errors="replace", timeout=400, env=dict(os.environ, FGPU_CLI_TIMES="1"))
                rc, err = r.returncode, r.stderr
            except subprocess.TimeoutExpired:
                rc, err = "cut off after 400 s", ""
            out[tag] = (rc, time.time() - t0, err)
        diff = []
        for fn in sorted(os.listdir(os.path.join(td, "gpu"))):
            a, b = os.path.join(td, "gpu", fn), os.path.join(td, "ref", fn)
            x = open(a, "rb").read()
            y = open(b, "rb").read() if os.path.exists(b) else None
            if y is None or (x != y and not (fn.endswith("short_pair_filter") and out["ref"][0] != 0 and x[:len(y)] == y)):
                diff.append((fn, len(x), None if y is None else len(y)))
        passes = re.findall(r"pass ([12]) \([^)]*\)\s+([0-9.]+) ms", out["gpu"][2])
        print(f"{name} ({len(lines)} reads, k = {k}): {'equal' if not diff else 'DIFFERENT ' + str(diff)} | command line {out['gpu'][1]:.1f} s (exit {out['gpu'][0]}; passes {passes}), "
              f"reference {out['ref'][1]:.1f} s (exit {out['ref'][0]})", flush=True)
        bad += 1 if diff else 0
print("failures:", bad)
sys.exit(1 if bad else 0)
