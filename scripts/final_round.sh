#!/bin/bash
# ON THE GPU BOX: the round's committed evidence in one call -- the driver's bench command, rocprofv3 kernel statistics + the two PMC passes of
# the same command (scripts/profile_round.sh), the strong-scaling projection, the command line at the size of configs 5 and 4.
#   gpurun --timeout 1150 -- 'bash scripts/final_round.sh r06'
tag=${1:-r06}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/final_$tag
mkdir -p "$out"
cd "$root"
python3 bench.py --steps 20 --warmup 5 > "$out/bench.json" 2> "$out/bench.err" || { tail -5 "$out/bench.err"; exit 1; }
echo "bench: $(cut -c1-200 "$out/bench.json")"
bash scripts/profile_round.sh "$tag" > "$out/profile_round.log" 2>&1 || { tail -5 "$out/profile_round.log"; exit 1; }
echo "profiles done"
python3 scripts/project_strong.py 2 4 8 > "$out/project_strong.txt" 2> "$out/project_strong.err" || { tail -5 "$out/project_strong.err"; exit 1; }
tail -4 "$out/project_strong.txt" | cut -c1-420
for c in config5 config4; do
  python3 scripts/cli_large.py $c > "$out/cli_large_$c.json" 2> "$out/cli_large_$c.err" || { tail -5 "$out/cli_large_$c.err"; exit 1; }
  echo "cli $c: $(cut -c1-260 "$out/cli_large_$c.json")"
done
