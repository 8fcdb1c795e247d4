#!/bin/bash
# HIP API time of one CLI run on config 2's size (measurement aid; run on the GPU box after scripts/cli_e2e.py has left /tmp/e2e_reads.fa)
set -e
out=gpurun_out/cli_trace
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
n=${1:-10000000}
FGPU_CLI_TIDY=1 rocprofv3 --hip-trace --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out -o cli -- $GRAFT_REPO_ROOT/faucet_amd/faucet -read_load_file /tmp/e2e_reads.fa -read_scan_file /tmp/e2e_reads.fa \
  -size_kmer 31 -max_read_length 100 -estimated_kmers $((10 * n)) -singletons $((2 * n)) --no_cleaning -file_prefix /tmp/e2e_trace -chunk_mb 64 > $GRAFT_REPO_ROOT/$out/stdout.txt 2> $GRAFT_REPO_ROOT/$out/stderr.txt
cd $GRAFT_REPO_ROOT
find $out -name "*hip_api_stats.csv" | head -1 | xargs -I{} sh -c 'head -25 {}'
