#!/bin/bash
# ON THE GPU BOX: config 3's shape through the CLI N times with the window-size controller's trace (FGPU_DEBUG_SPAN): pass times and, per run, the
# decisions.  bash scripts/config3_span_trace.sh [N] [path of the tree to use]
n=${1:-6}
root=${GRAFT_REPO_ROOT:-$(pwd)}
tree=${2:-$root}
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
PY
for i in $(seq $n); do
  FGPU_DEBUG_SPAN=1 FGPU_CLI_TIMES=1 $tree/faucet_amd/faucet -read_load_file /dev/shm/c3_reads.fq -read_scan_file /dev/shm/c3_reads.fq -file_prefix /dev/shm/c3_out $(cat /dev/shm/c3_args.txt) > /dev/null 2> /tmp/c3_$i.err
  echo "run $i: $(grep -E 'pass 1 \(|pass 2 \(' /tmp/c3_$i.err | awk '{printf "%s %s ms  ", $2$3, $(NF-4)}') decisions: $(grep -c '\[span\]' /tmp/c3_$i.err)"
  grep '\[span\]' /tmp/c3_$i.err | awk '{printf "      %s\n", $0}' | cut -c1-140
done
rm -f /dev/shm/c3_reads.fq /dev/shm/c3_out.* /dev/shm/c3_args.txt
