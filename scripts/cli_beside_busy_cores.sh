#!/bin/bash
# ON THE GPU BOX: config 3's shape through the command line while N processes of this job keep cores busy (what other work on the host would do to a
# pass that waits for the device some 120 times): pass times by the CLI's own clock, N = 0, 8, 16, 32, 64.
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
nproc
cat /sys/fs/cgroup/cpu.max 2>/dev/null
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
for n in 0 8 16 32 64 0; do
  pids=""
  for i in $(seq 1 $n); do ( while :; do :; done ) & pids="$pids $!"; done
  sleep 0.5
  for rep in 1 2; do
    s=$(date +%s%N)
    FGPU_CLI_TIMES=1 $root/faucet_amd/faucet -read_load_file /dev/shm/c3_reads.fq -read_scan_file /dev/shm/c3_reads.fq -file_prefix /dev/shm/c3_out $(cat /dev/shm/c3_args.txt) > /dev/null 2> /tmp/c3.err
    e=$(date +%s%N)
    echo "$n busy processes: process $(( (e - s) / 1000000 )) ms  $(grep -E 'pass 1 \(|pass 2 \(|fgpu_create ' /tmp/c3.err | awk '{printf "%s %s ms  ", $2$3, $(NF-4)}')"
  done
  for p in $pids; do kill $p 2>/dev/null; done
  wait 2>/dev/null
done
rm -f /dev/shm/c3_reads.fq /dev/shm/c3_out.* /dev/shm/c3_args.txt
