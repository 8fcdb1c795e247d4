#!/bin/bash
# Runs ON THE GPU BOX: BASELINE config 3's shape through the CLI, without a profiler, N times: the CLI's own phase clock (FGPU_CLI_TIMES) and the digests
# of the four files.  Extra VAR=value arguments are exported first.     gpurun -- 'bash scripts/config3_cli_times.sh 3 [VAR=val ...]'
n=${1:-3}; shift
for kv in "$@"; do export "$kv"; done
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
python3 - <<'PY'
import json, os, sys
sys.path.insert(0, os.getcwd())
import torch
from faucet_amd import synth_det as sd
fx = json.load(open("tests/golden/fullsize.json"))["config3"]
c = fx["params"]
dev = torch.device("cuda", 0)
g = sd.make_genome(c["genome"], c["genome_seed"], dev)
sd.plant_repeats(g, c["genome_seed"] + 100, *c["repeats"])
reads = sd.make_pairs(g, c["pairs"], c["read_len"], c["insert"][0], c["insert"][1], c["err"], c["read_seed"], dev)
sd.fasta_bytes(reads, fastq=True).cpu().numpy().tofile("/dev/shm/c3_reads.fq")
open("/dev/shm/c3_args.txt", "w").write(" ".join(fx["args"]))
print("expected", fx["junctions_sha256"][:16], fx["long_pair_filter_sha256"][:16], fx["short_pair_filter_sha256"][:16], fx["bloom_sha256"][:16])
PY
for i in $(seq $n); do
  s=$(date +%s%N)
  FGPU_CLI_TIMES=1 $root/faucet_amd/faucet -read_load_file /dev/shm/c3_reads.fq -read_scan_file /dev/shm/c3_reads.fq -file_prefix /dev/shm/c3_out $(cat /dev/shm/c3_args.txt) > /dev/null 2> /tmp/c3.err
  e=$(date +%s%N)
  echo "run $i: process $(( (e - s) / 1000000 )) ms  $(grep -E 'pass 1 \(|pass 2 \(' /tmp/c3.err | awk '{printf "%s %s ms  ", $2$3, $(NF-4)}')"
  grep -E "optimistically|long pair filter" /tmp/c3.err | sed 's/^\[cli\] */    /'
done
echo "got      $(sha256sum /dev/shm/c3_out.junctions /dev/shm/c3_out.long_pair_filter /dev/shm/c3_out.short_pair_filter /dev/shm/c3_out.bloom | cut -c1-16 | tr '\n' ' ')"
rm -f /dev/shm/c3_reads.fq /dev/shm/c3_out.* /dev/shm/c3_args.txt
