"""File-to-files wall time of the `faucet` command line on a config-2-sized FASTA (diagnostic; run on the GPU box).

    PYTHONPATH=. python scripts/cli_e2e.py [n_reads]

Writes a synthetic FASTA (fixed-width headers), runs the CLI with the records split on the host (-batch_reads) and on
the device (default), and prints the wall time of each run and the output sizes.  Both runs must write the same files.
"""
import hashlib
import os
import subprocess
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (make_genome / make_reads: the bench's generators)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
fastq = len(sys.argv) > 2 and sys.argv[2] == "fastq"
L, G = 100, 2 * n
dev = torch.device("cuda", 0)
reads = bench.make_reads(bench.make_genome(G, 2, dev), n, L, 0.01, 1000, dev).cpu().numpy()
rec = np.empty((n, 10 + L + 1 + ((2 + L + 1) if fastq else 0)), dtype=np.uint8)
rec[:, 0] = ord("@") if fastq else ord(">")
idx = np.arange(n, dtype=np.int64)
for d in range(8):
    rec[:, 8 - d] = ord("0") + (idx // 10 ** d) % 10
rec[:, 9] = ord("\n")
rec[:, 10:10 + L] = reads
rec[:, 10 + L] = ord("\n")
if fastq:
    rec[:, 11 + L] = ord("+")
    rec[:, 12 + L] = ord("\n")
    rec[:, 13 + L:13 + 2 * L] = ord("I")
    rec[:, 13 + 2 * L] = ord("\n")
path = "/tmp/e2e_reads.fq" if fastq else "/tmp/e2e_reads.fa"
rec.tofile(path)
print(f"{path}: {os.path.getsize(path) / 1e9:.2f} GB, {n} reads")
del rec, reads
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "faucet_amd", "faucet")
base = ["-read_load_file", path, "-read_scan_file", path, "-size_kmer", "31", "-max_read_length", str(L),
        "-estimated_kmers", str(10 * n), "-singletons", str(2 * n), "--no_cleaning"] + (["--fastq"] if fastq else [])


def digest(p):
    h = hashlib.sha256()
    with open(p, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()[:16]


sigs = []
for name, extra in (("host getline, 1 M reads per call", ["-batch_reads", "1000000"]), ("device split, 256 MB of text per call", []),
                    ("device split, 64 MB of text per call", ["-chunk_mb", "64"])):
    pref = "/tmp/e2e_" + name.split()[0] + str(len(sigs))
    t0 = time.perf_counter()
    r = subprocess.run([exe] + base + ["-file_prefix", pref] + extra, capture_output=True, text=True)
    dt = time.perf_counter() - t0
    assert r.returncode == 0, r.stderr[-2000:]
    sig = (digest(pref + ".bloom"), digest(pref + ".junctions"))
    sigs.append(sig)
    print(f"{name:42s} {dt:7.2f} s  {70 * n / dt / 1e6:8.1f} M k-mers/s file to files   bloom {sig[0]} junctions {sig[1]}")
    for line in r.stderr.splitlines():          # FGPU_CLI_TIMES=1: the CLI's own phase clock
        if line.startswith("[cli]"):
            print("    " + line)
assert len(set(sigs)) == 1, "the runs disagree"
