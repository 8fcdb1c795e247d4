#!/bin/bash
# Measurement build of the CLI (-DFGPU_CLI_PROFILE: tick counters inside the pair-filter worker's long-pair loop) on BASELINE config 3's shape.  GPU box.
set -e
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
g++ -std=c++11 -O2 -DFGPU_CLI_PROFILE -I include faucet_amd/host/faucet_main.cpp -o /tmp/faucet_prof -L faucet_amd -lfaucet_gpu -Wl,-rpath,$root/faucet_amd -Wl,-rpath-link,/opt/rocm/lib -lpthread
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
set +e
FGPU_CLI_TIMES=1 timeout -k 10 120 /tmp/faucet_prof -read_load_file /dev/shm/c3_reads.fq -read_scan_file /dev/shm/c3_reads.fq -file_prefix /dev/shm/c3_out $(cat /dev/shm/c3_args.txt) 2>&1 >/dev/null | grep -E "applying|preparing|cli-profile|pass 2"
rm -f /dev/shm/c3_reads.fq /dev/shm/c3_out.* /dev/shm/c3_args.txt
grep -m1 "model name" /proc/cpuinfo; grep -m1 "cpu MHz" /proc/cpuinfo
