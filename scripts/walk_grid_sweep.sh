#!/bin/bash
# Grid shapes of the walk stage's bookkeeping on the large configurations: the delta's share of k_walk_register's grid (FGPU_DELTA_PER_BLOCK = created keys
# per block of 256 threads; unset = 64 blocks flat) and k_walk_cluster's grid cap (FGPU_CLUSTER_GRID, default 256).  Config 2's step + full-size legs.
root=${GRAFT_REPO_ROOT:-$(pwd)}
run() {
  echo "=== $*"
  env "$@" python3 "$root/bench.py" --steps 10 --warmup 3 --no-cpu --no-host-leg --no-ceilings --no-full-size | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step_rank0']
print('config 2: step', round(d['ms_per_step'],2), 'ms; walk_stage', k.get('walk_stage'))"
  for c in ${CONFIGS:-config5 config4}; do
    env "$@" python3 "$root/scripts/fullsize_step.py" $c | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$c:', [ (round(s['seconds'],3), s['counters_equal_the_oracles'], s['kernel_ms'].get('walk_stage')) for s in (d['first_step_of_the_context'], d['second_step'])])"
  done
}
if [ -n "$RUNS" ]; then
  IFS=';' read -ra L <<< "$RUNS"
  for r in "${L[@]}"; do run $r; done
else
  run FGPU_NOP=1
  run FGPU_DELTA_PER_BLOCK=4096
  run FGPU_DELTA_PER_BLOCK=1024
  run FGPU_DELTA_PER_BLOCK=1024 FGPU_CLUSTER_GRID=1024
  run FGPU_CLUSTER_GRID=1024
fi
