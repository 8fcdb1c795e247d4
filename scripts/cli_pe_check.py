"""Paired-end FASTQ at scale through the CLI (diagnostic; GPU box): host getline batches vs device-side splitting must write the same
four files (.bloom, .junctions, .short_pair_filter, .long_pair_filter)."""
import hashlib
import os
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faucet_amd import synth  # noqa: E402

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1_500_000
g = synth.make_genome(6_000_000, 31, repeats=20, repeat_len=400)
r = synth.make_pairs(g, n_pairs, 100, 300, 30, 0.01, 32)
n, L = r.shape
rec = np.empty((n, 10 + L + 1 + 2 + L + 1), dtype=np.uint8)
rec[:, 0] = ord("@")
idx = np.arange(n, dtype=np.int64)
for d in range(8):
    rec[:, 8 - d] = ord("0") + (idx // 10 ** d) % 10
rec[:, 9] = ord("\n")
rec[:, 10:10 + L] = r
rec[:, 10 + L] = ord("\n")
rec[:, 11 + L] = ord("+")
rec[:, 12 + L] = ord("\n")
rec[:, 13 + L:13 + 2 * L] = ord("I")
rec[:, 13 + 2 * L] = ord("\n")
path = "/tmp/pe_reads.fq"
rec.tofile(path)
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "faucet_amd", "faucet")
base = ["-read_load_file", path, "-read_scan_file", path, "-size_kmer", "31", "-max_read_length", "100", "-estimated_kmers", str(8 * n),
        "-singletons", str(2 * n), "--fastq", "--paired_ends"]


def digest(p):
    h = hashlib.sha256()
    with open(p, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()[:12]


sigs = []
for name, extra in (("host getline 333333 reads per call", ["-batch_reads", "333333"]), ("device split 64 MB", []), ("device split 7 MB", ["-chunk_mb", "7"])):
    pref = "/tmp/pe_" + str(len(sigs))
    t0 = time.perf_counter()
    p = subprocess.run([exe] + base + ["-file_prefix", pref] + extra, capture_output=True, text=True)
    dt = time.perf_counter() - t0
    assert p.returncode == 3, p.stderr[-1500:]
    sig = tuple(digest(pref + e) for e in (".bloom", ".junctions", ".short_pair_filter", ".long_pair_filter"))
    sigs.append(sig)
    counts = [ln for ln in p.stdout.splitlines() if "Empty count" in ln]
    print(f"{name:36s} {dt:6.2f} s  {sig}  {counts[-1] if counts else ''}", flush=True)
    for line in p.stderr.splitlines():          # FGPU_CLI_TIMES=1: the CLI's own phase clock
        if line.startswith("[cli]"):
            print("    " + line)
print("PASS" if len(set(sigs)) == 1 else "FAIL")
