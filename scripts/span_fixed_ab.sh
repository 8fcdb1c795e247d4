#!/bin/bash
# ON THE GPU BOX: config 2's step with the walk's window size fixed (FAUCET_WALK_SPAN) against the controller (0)
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$root"
for s in 0 $((1<<23)) $((1<<24)) $((3<<23)) $((1<<25)) $((3<<24)) $((1<<26)) 0; do
  FAUCET_WALK_SPAN=$s python3 bench.py --steps 20 --warmup 5 --no-cpu --no-host-leg --no-full-size --no-profile 2> /dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('FAUCET_WALK_SPAN=$s', round(d['ms_per_step'],2), 'ms/step', 'windows', d['outputs']['walk_windows_rank0'], 'followers', d['outputs']['walk_followers_rank0'], 'walk_stage', d['kernel_ms_per_step_rank0'].get('walk_stage'))"
done
