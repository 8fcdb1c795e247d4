#!/bin/bash
# ON THE GPU BOX: the scan's batch-buffer depth (how many batches the pure stage may run ahead of the walk), config 2's step, 3 runs each
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$root"
for d in 2 3 4 2 3 4; do
  FGPU_SCAN_BUFFERS=$d python3 bench.py --steps 20 --warmup 5 --no-cpu --no-host-leg --no-full-size --no-profile 2> /dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('FGPU_SCAN_BUFFERS=$d', round(d['ms_per_step'],2), 'ms/step', 'junctions', d['outputs']['junctions'], 'fallbacks', d['lazy_flag_fallbacks'], 'filled_in_walk', d['outputs'].get('flags_filled_in_walk_rank0'))"
done
