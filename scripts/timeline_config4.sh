#!/bin/bash
# ON THE GPU BOX: kernel trace of BASELINE config 4's whole workload on one GPU (warm-up step + one step), the last step's timeline
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/timeline4
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d "$out/t" -o run -- python3 "$root/bench.py" --reads 200000000 --genome 400000000 --estimated-kmers 1000000000 --singletons 200000000 --batch-reads 2500000 --steps 1 --warmup 1 --no-cpu --no-ceilings --no-host-leg --no-profile --no-full-size > "$out/bench.json" 2> "$out/bench.err"
f=$(find "$out/t" -name "*kernel_trace.csv" | head -1)
python3 "$root/scripts/timeline.py" "$f" 0 x > "$out/timeline.txt" 2>&1
find "$out" \( -name "*kernel_trace.csv" -o -name "*.db" \) -delete
head -12 "$out/timeline.txt" | cut -c1-400
