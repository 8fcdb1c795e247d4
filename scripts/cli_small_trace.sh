#!/bin/bash
# HIP API time of one CLI run on config 1 (1 000 reads): where the 35 ms of a scan over 1 000 reads go (measurement aid; GPU box).
set -e
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/cli_small_trace
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
gunzip -c $root/tests/golden/c1_k21/reads.fa.gz > /tmp/c1_reads.fa
args=$(python3 -c "import json; print(' '.join(a for a in json.load(open('$root/tests/golden/c1_k21/case.json'))['args'] if not a.endswith('.fa')))")
# FGPU_CLI_TIDY=1: the CLI must leave through exit(), not _exit(), or the profiler never writes its files (and waits for ever)
FGPU_CLI_TIDY=1 timeout -k 10 120 rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d $out -o cli -- $root/faucet_amd/faucet -read_load_file /tmp/c1_reads.fa -read_scan_file /tmp/c1_reads.fa -file_prefix /tmp/c1_out $args > $out/stdout.txt 2> $out/stderr.txt
f=$(find $out -name "*hip_api_stats.csv" | head -1)
head -16 $f
python3 - <<PY
import csv, glob
f = glob.glob("$out/**/*hip_api_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
slow = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), (int(r["Start_Timestamp"]) - t0) / 1e6, r["Function"]) for r in rows]
print("calls over 0.3 ms, in time order (ms since the first API call):")
for d, at, fn in slow:
    if d > 300000:
        print(f"  {at:9.2f}  {d / 1e6:8.2f} ms  {fn}")
PY
