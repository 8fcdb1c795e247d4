#!/bin/bash
# Runs ON THE GPU BOX: instruction-cache counters of the key-ordered walk (and of k_walk) on scripts/pe_profile.py's data.
tag=$1; shift 1
for kv in "$@"; do export "$kv"; done
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --kernel-include-regex "k_walk" --output-format csv -d "$out/ic" -o run -- python3 "$root/scripts/pe_profile.py" > "$out/p1.log" 2> "$out/p1.err"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_INSTS_SALU SQ_INSTS_VALU --kernel-include-regex "k_walk" --output-format csv -d "$out/sq" -o run -- python3 "$root/scripts/pe_profile.py" > "$out/p2.log" 2> "$out/p2.err"
find "$out" \( -name "*.db" \) -delete
python3 - "$out" <<'PY'
import csv, glob, os, sys
acc = {}
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][-40:]
        a = acc.setdefault((k, row["Counter_Name"]), [0, 0.0])
        a[0] += 1; a[1] += float(row["Counter_Value"])
for (k, c), (n, v) in sorted(acc.items()):
    print(f"{k:42s} {c:28s} launches {n:5d} total {v:.4g} per-launch {v/n:.4g}")
PY
