#!/bin/bash
# Runs ON THE GPU BOX: SQ counters of one kernel (regex $2) of a short bench run with the bench arguments given in $3 (one string).
#   gpurun -- 'bash scripts/pmc_kernel_args.sh tag "k_walk\\(" "--reads 25000000 --genome 400000000 ..."'
tag=$1; regex=$2; args=$3
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-include-regex "$regex" --output-format csv -d "$out/sq1" -o run -- python3 "$root/bench.py" --steps 1 --warmup 0 --no-cpu --no-ceilings --no-host-leg --no-profile $args > "$out/b1.json" 2> "$out/b1.err"
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --kernel-include-regex "$regex" --output-format csv -d "$out/sq2" -o run -- python3 "$root/bench.py" --steps 1 --warmup 0 --no-cpu --no-ceilings --no-host-leg --no-profile $args > "$out/b2.json" 2> "$out/b2.err"
rocprofv3 --kernel-trace --stats --kernel-include-regex "$regex" --output-format csv -d "$out/st" -o run -- python3 "$root/bench.py" --steps 1 --warmup 0 --no-cpu --no-ceilings --no-host-leg --no-profile $args > "$out/b3.json" 2> "$out/b3.err"
find "$out" \( -name "*.db" -o -name "*kernel_trace.csv" \) -delete
python3 - "$out" <<'PY'
import csv, glob, os, sys
acc = {}
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][-40:]
        a = acc.setdefault((k, row["Counter_Name"]), [0, 0.0])
        a[0] += 1; a[1] += float(row["Counter_Value"])
for (k, c), (n, v) in sorted(acc.items()):
    print(f"{k:42s} {c:24s} launches {n:5d} total {v:.4g} per-launch {v/n:.4g}")
for f in glob.glob(os.path.join(sys.argv[1], "st", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        print("stats", row["Name"][:60], row["Calls"], row["TotalDurationNs"], row["AverageNs"])
PY
