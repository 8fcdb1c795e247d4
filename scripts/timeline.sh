#!/bin/bash
# ON THE GPU BOX: kernel trace of three steps of bench.py, the last step's timeline (scripts/timeline.py)
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/timeline
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d "$out/t" -o run -- python3 "$root/bench.py" --steps 3 --warmup 2 --no-cpu --no-host-leg --no-full-size --no-profile > "$out/bench.json" 2> "$out/bench.err"
f=$(find "$out/t" -name "*kernel_trace.csv" | head -1)
head -1 "$f" > "$out/header.txt"
python3 "$root/scripts/timeline.py" "$f" 1 ${DETAIL:+x} > "$out/timeline.txt" 2>&1
find "$out" \( -name "*kernel_trace.csv" -o -name "*.db" \) -delete
tail -c 600 "$out/bench.json"
