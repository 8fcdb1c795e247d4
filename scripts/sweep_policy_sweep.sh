#!/bin/bash
# When does a sweep of first[] pay?  (load.hip: after a batch, once the epoch has grown to FGPU_SWEEP_RATIO of what the carry covers AND holds tai / FGPU_SWEEP_MIN_FRAC accesses.)
# One rank's own load of config 4 on 8 GPUs (scripts/shard_load_times.py 8: the line "own load with shard times"), config 2's step, configs 5 and 4 whole.
root=${GRAFT_REPO_ROOT:-$(pwd)}
run() {
  echo "=== $*"
  env "$@" python3 "$root/scripts/shard_load_times.py" 8 2>/dev/null | grep "own load with shard times run 1\|load keep_carry=False run 2" | cut -c1-200
  env "$@" python3 "$root/bench.py" --steps 10 --warmup 3 --no-cpu --no-host-leg --no-ceilings --no-full-size | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step_rank0']
print('config 2: step', round(d['ms_per_step'],2), 'ms; load_mark', k.get('load_mark'), 'carry_update', k.get('carry_update'), 'load_resolve', k.get('load_resolve'))"
  for c in ${CONFIGS:-config5 config4}; do
    env "$@" python3 "$root/scripts/fullsize_step.py" $c | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$c:', [ (round(s['seconds'],3), s['counters_equal_the_oracles'], s['kernel_ms'].get('load_mark'), s['kernel_ms'].get('carry_update')) for s in (d['first_step_of_the_context'], d['second_step'])])"
  done
}
if [ -n "$RUNS" ]; then
  IFS=';' read -ra L <<< "$RUNS"
  for r in "${L[@]}"; do run $r; done
else
  run FGPU_NOP=1
  run FGPU_SWEEP_MIN_FRAC=16
  run FGPU_SWEEP_MIN_FRAC=4
  run FGPU_SWEEP_MIN_FRAC=16 FGPU_SWEEP_RATIO=2/1
fi
