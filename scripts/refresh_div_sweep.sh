#!/bin/bash
# When is it cheaper to make a window's snapshot planes again than to register the keys created since they were made?  (scan_walk.hip, refresh_window:
# FGPU_REFRESH_DIV = d: planes again when the estimated delta exceeds positions / d.)  Config 2's step and the full-size legs of configs 5 and 4.
root=${GRAFT_REPO_ROOT:-$(pwd)}
for d in ${DIVS:-4 16 48 128}; do
  echo "=== FGPU_REFRESH_DIV=$d"
  FGPU_REFRESH_DIV=$d python3 "$root/bench.py" --steps 10 --warmup 3 --no-cpu --no-host-leg --no-ceilings --no-full-size | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step_rank0']
print('config 2: step', round(d['ms_per_step'],2), 'ms; walk_stage', k.get('walk_stage'), 'need_lookup', k.get('need_lookup'))"
  for c in config5 config4; do
    FGPU_REFRESH_DIV=$d python3 "$root/scripts/fullsize_step.py" $c | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$c:', [ (round(s['seconds'],3), s['counters_equal_the_oracles'], s['kernel_ms'].get('walk_stage')) for s in (d['first_step_of_the_context'], d['second_step'])])"
  done
done
