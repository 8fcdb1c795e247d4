#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 kernel statistics and the two PMC passes of bench.py.
#   gpurun --timeout 1500 -- 'bash scripts/profile_round.sh r01'
# Outputs land under gpurun_out/prof_<tag>/; scripts/pmc_summary.py turns them into the files kept under profiles/.
# The program itself follows `--` (never a shell or env wrapper), and --pmc is never combined with a trace domain.
tag=${1:-r01}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o run -- python3 "$root/bench.py" --steps 5 --warmup 2 --no-cpu --no-host-leg > "$out/bench_stats.json" 2> "$out/bench_stats.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -o run -- python3 "$root/bench.py" --steps 1 --warmup 0 --no-cpu --no-host-leg > "$out/bench_fetch.json" 2> "$out/bench_fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -o run -- python3 "$root/bench.py" --steps 1 --warmup 0 --no-cpu --no-host-leg > "$out/bench_write.json" 2> "$out/bench_write.err"
# the per-dispatch trace is large; the statistics and the counter tables are what is kept
find "$out" \( -name "*kernel_trace.csv" -o -name "*.db" \) -delete
du -sh "$out"
ls -R "$out" | head -40
