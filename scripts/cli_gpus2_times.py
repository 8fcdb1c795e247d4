"""`faucet -gpus N` on config 2's reads as a FASTA file (all shards on this box's one device), FGPU_CLI_TIMES=1: per-rank stage times, with the
walk merging the new keys into a prepared shard's planes (default) and with the planes made again in full (FGPU_NO_DELTA_REFRESH=1).
    python scripts/cli_gpus2_times.py [gpus = 2] [runs = 3]"""
import os, re, subprocess, sys, tempfile, time, shutil
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

gpus = sys.argv[1] if len(sys.argv) > 1 else "2"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
n, L_ = 10_000_000, 100
reads = bench.make_reads(bench.make_genome(20_000_000, 2, dev), n, L_, 0.01, 1000, dev).cpu().numpy()
d = tempfile.mkdtemp(prefix="faucet_g2_", dir="/dev/shm")
try:
    rec = np.empty((n, 10 + L_ + 1), dtype=np.uint8)
    rec[:, 0] = ord(">")
    idx = np.arange(n, dtype=np.int64)
    for dgt in range(8):
        rec[:, 8 - dgt] = ord("0") + (idx // 10 ** dgt) % 10
    rec[:, 9] = ord("\n"); rec[:, 10:10 + L_] = reads; rec[:, 10 + L_] = ord("\n")
    path = os.path.join(d, "reads.fa"); rec.tofile(path); del rec
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "faucet_amd", "faucet")
    cmd = [exe, "-read_load_file", path, "-read_scan_file", path, "-size_kmer", "31", "-max_read_length", "100", "-estimated_kmers", "100000000",
           "-singletons", "20000000", "--no_cleaning", "-file_prefix", os.path.join(d, "out"), "-gpus", gpus]
    for env in ({}, {"FGPU_NO_DELTA_REFRESH": "1"}) * runs:
        t0 = time.perf_counter()
        r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, FGPU_CLI_TIMES="1", **env))
        dt = time.perf_counter() - t0
        p2 = [ln.strip() for ln in r.stderr.splitlines() if "pass 2:" in ln or "pass 2 (" in ln]
        print(("planes made again " if env else "new keys merged   "), f"process {dt * 1e3:.0f} ms |", " | ".join(x[:110] for x in p2), flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
