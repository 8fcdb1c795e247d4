#!/bin/bash
# ON THE GPU BOX: kernel trace of one hop of config 4's chain (scripts/hop_kernels.py N HOP), cut to the hop by scripts/hop_timeline.py
#   bash scripts/hop_timeline.sh [N=8] [HOP=2] [tag]
root=${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-8}; HOP=${2:-2}; tag=${3:-hop}
out=$root/gpurun_out/$tag
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 700 rocprofv3 --kernel-trace --output-format csv -d "$out/t" -o run -- python3 "$root/scripts/hop_kernels.py" "$N" "$HOP" > "$out/hop_kernels.txt" 2> "$out/hop_kernels.err" || { tail -5 "$out/hop_kernels.err"; exit 1; }
python3 "$root/scripts/hop_timeline.py" "$out/t" > "$out/timeline.txt" 2>&1
find "$out" \( -name "*kernel_trace.csv" -o -name "*.db" \) -delete
cat "$out/hop_kernels.txt" | cut -c1-600
cat "$out/timeline.txt"
