#!/bin/bash
# ON THE GPU BOX: config 2's reads as a FASTA file in /dev/shm through the `faucet` command line, N times, with its phase clock (FGPU_CLI_TIMES=1) and
# the library's own account of fgpu_create: where the 0.6-0.8 s of a file-to-files run go.  Extra VAR=value arguments are exported first.
n=${1:-3}; shift
for kv in "$@"; do export "$kv"; done
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
python3 - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch, bench
dev = torch.device("cuda", 0)
n, L = 10_000_000, 100
reads = bench.make_reads(bench.make_genome(20_000_000, 2, dev), n, L, 0.01, 1000, dev).cpu().numpy()
rec = np.empty((n, 10 + L + 1), dtype=np.uint8)
rec[:, 0] = ord(">")
idx = np.arange(n, dtype=np.int64)
for d in range(8):
    rec[:, 8 - d] = ord("0") + (idx // 10 ** d) % 10
rec[:, 9] = ord("\n"); rec[:, 10:10 + L] = reads; rec[:, 10 + L] = ord("\n")
rec.tofile("/dev/shm/c2_reads.fa")
PY
for i in $(seq $n); do
  s=$(date +%s%N)
  FGPU_CLI_TIMES=1 $root/faucet_amd/faucet -read_load_file /dev/shm/c2_reads.fa -read_scan_file /dev/shm/c2_reads.fa -file_prefix /dev/shm/c2_out -size_kmer 31 -max_read_length 100 -estimated_kmers 100000000 -singletons 20000000 --no_cleaning $EXTRA > /dev/null 2> /tmp/c2.err
  e=$(date +%s%N)
  echo "=== run $i: process $(( (e - s) / 1000000 )) ms"
  grep -E "^\[cli\]|^\[fgpu_create\]" /tmp/c2.err
done
rm -f /dev/shm/c2_reads.fa /dev/shm/c2_out.*
