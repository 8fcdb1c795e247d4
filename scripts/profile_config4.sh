#!/bin/bash
# rocprofv3 kernel statistics of BASELINE config 4's whole workload on one GPU (one bench step; GPU box).  Output under gpurun_out/prof_config4/.
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_config4
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o run -- python3 "$root/bench.py" --reads 200000000 --genome 400000000 --estimated-kmers 1000000000 --singletons 200000000 --batch-reads 2500000 --steps 1 --warmup 1 --no-cpu --no-ceilings --no-host-leg > "$out/bench.json" 2> "$out/bench.err"
echo "rc=$?"
find "$out" \( -name "*kernel_trace.csv" -o -name "*.db" \) -delete
python3 - <<PY
import csv, glob
f = glob.glob("$out/**/*kernel_stats.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_" in r["Name"] and "at::" not in r["Name"] and "rocprim" not in r["Name"]]
w = csv.writer(open("$out/r04_rocprofv3_kernel_stats_config4.csv", "w"))
w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
for r in rows:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
    w.writerow([n, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    print(f"{n[:40]:40s} x{r['Calls']:>6s} {int(r['TotalDurationNs'])/1e6:10.1f} ms")
PY
