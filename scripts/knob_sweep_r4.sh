#!/bin/bash
# ON THE GPU BOX: a few launch-shape knobs on config 2's step after round 4's controller change (each value twice, interleaved)
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$root"
run() { env "$@" python3 bench.py --steps 20 --warmup 5 --no-cpu --no-host-leg --no-full-size --no-profile 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-34s %.2f ms/step' % ('$*', d['ms_per_step']))"; }
for rep in 1 2; do
  run FGPU_NONE=1
  run FGPU_WALK_DYN_GRID=512
  run FGPU_WALK_DYN_GRID=2048
  run FGPU_FLAGS_SM_BLOCKS=2048
  run FGPU_FLAGS_SM_BLOCKS=8192
  run FGPU_RESOLVE_SM=2048
  run FGPU_RESOLVE_SM=8192
done
