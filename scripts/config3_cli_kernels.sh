#!/bin/bash
# Kernel time of the faucet CLI on BASELINE config 3's shape (2.5 M pairs, --fastq --paired_ends, cleaning on): what pass 2's ~0.9 s are made of.
# GPU box; measurement aid.  FGPU_CLI_TIDY=1: the CLI leaves through exit() so that the profiler writes its files.
set -e
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/config3_cli_kernels
mkdir -p $out
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
cd /tmp && export TMPDIR=/tmp
set +e
FGPU_CLI_TIDY=1 FGPU_CLI_TIMES=1 timeout -k 10 180 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o cli -- $root/faucet_amd/faucet -read_load_file /dev/shm/c3_reads.fq -read_scan_file /dev/shm/c3_reads.fq -file_prefix /dev/shm/c3_out $(cat /dev/shm/c3_args.txt) > $out/stdout.txt 2> $out/stderr.txt
echo "faucet rc=$? (3 = outputs written, contig graph not built)"
rm -f /dev/shm/c3_reads.fq /dev/shm/c3_out.* /dev/shm/c3_args.txt
grep "^\[cli\]" $out/stderr.txt | head -20
python3 - <<PY
import csv, glob
f = glob.glob("$out/**/*kernel_stats.csv", recursive=True)[0]
print("kernels by total time:")
for r in list(csv.DictReader(open(f)))[:14]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print(f"  {n[:50]:50s} x{r['Calls']:>5s} {int(r['TotalDurationNs']) / 1e6:9.2f} ms")
PY
find $out -name "*kernel_trace.csv" -delete
