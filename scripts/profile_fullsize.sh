#!/bin/bash
# rocprofv3 kernel statistics of one of BASELINE's full-size configurations on one GPU (bench.py's full-size leg: two steps; GPU box).
#   gpurun --timeout 900 -- 'bash scripts/profile_fullsize.sh config5 r05'      -> gpurun_out/prof_<config>/<tag>_rocprofv3_kernel_stats_<config>.csv
cfg=${1:-config5}
tag=${2:-r05}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$cfg
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
if [ "${3:-}" = "pmc" ]; then
  # round 6: the two counter passes of the same program, each in its own run (never with a trace domain); the summary goes into
  # profiles/pmc_traffic.json under full_size.<config> (scripts/pmc_fullsize_summary.py)
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 800 rocprofv3 --pmc $c --output-format csv -d "$out/pmc_$c" -o run -- python3 "$root/scripts/fullsize_step.py" "$cfg" > "$out/step_$c.json" 2> "$out/step_$c.err"
    echo "$c rc=$?"
  done
  python3 "$root/scripts/pmc_fullsize_summary.py" "$out" "$cfg"
  find "$out" \( -name "*counter_collection.csv" -o -name "*.db" \) -delete
  exit 0
fi
timeout -k 10 700 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o run -- python3 "$root/scripts/fullsize_step.py" "$cfg" > "$out/step.json" 2> "$out/step.err"
echo "rc=$?"
find "$out" \( -name "*kernel_trace.csv" -o -name "*.db" \) -delete
python3 - <<PY
import csv, glob
f = glob.glob("$out/**/*kernel_stats.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_" in r["Name"] and "at::" not in r["Name"] and "rocprim" not in r["Name"]]
w = csv.writer(open("$out/${tag}_rocprofv3_kernel_stats_${cfg}.csv", "w"))
w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
for r in rows:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
    w.writerow([n, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    print(f"{n[:40]:40s} x{r['Calls']:>6s} {int(r['TotalDurationNs'])/1e6:10.1f} ms")
PY
