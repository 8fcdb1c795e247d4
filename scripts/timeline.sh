#!/bin/bash
# Runs ON THE GPU BOX: kernel trace (start/end per dispatch, queue id) of a short bench run, summarised by scripts/timeline_summary.py
tag=$1; shift
for kv in "$@"; do export "$kv"; done
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/tl_$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d "$out/trace" -o run -- python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu --no-ceilings > "$out/bench.json" 2> "$out/bench.err"
python3 "$root/scripts/timeline_summary.py" "$out/trace" > "$out/summary.txt" 2>&1
find "$out" \( -name "*.db" -o -name "*kernel_trace.csv" \) -delete
cat "$out/summary.txt"
