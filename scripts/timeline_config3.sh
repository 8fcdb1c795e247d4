#!/bin/bash
# ON THE GPU BOX: kernel trace of BASELINE config 3's shape through the CLI, the run's timeline (scripts/timeline.py)
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/timeline3
rm -rf "$out"; mkdir -p "$out"
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
export TMPDIR=/tmp
cd /tmp
FGPU_CLI_TIDY=1 FGPU_CLI_TIMES=1 rocprofv3 --kernel-trace --output-format csv -d "$out/t" -o run -- $root/faucet_amd/faucet -read_load_file /dev/shm/c3_reads.fq -read_scan_file /dev/shm/c3_reads.fq -file_prefix /dev/shm/c3_out $(cat /dev/shm/c3_args.txt) > /dev/null 2> "$out/cli.err"
f=$(find "$out/t" -name "*kernel_trace.csv" | head -1)
python3 "$root/scripts/timeline.py" "$f" 0 x > "$out/timeline.txt" 2>&1
find "$out" \( -name "*kernel_trace.csv" -o -name "*.db" \) -delete
rm -f /dev/shm/c3_reads.fq /dev/shm/c3_out.* /dev/shm/c3_args.txt
grep -E "pass [12] \(" "$out/cli.err"
head -14 "$out/timeline.txt" | cut -c1-420
