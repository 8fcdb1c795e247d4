#!/bin/bash
# ON THE GPU BOX: config 2's step by batch size and ramp (how the caller cuts the reads into batches)
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$root"
for cfg in "1000000 2" "1000000 0" "1000000 1" "1000000 3" "500000 2" "2000000 2" "2000000 3" "1000000 2"; do
  set -- $cfg
  python3 bench.py --steps 20 --warmup 5 --no-cpu --no-host-leg --no-full-size --no-profile --batch-reads $1 --ramp $2 2> /dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('batch $1 ramp $2:', round(d['ms_per_step'],2), 'ms/step', 'windows', d['outputs']['walk_windows_rank0'], 'filled_in_walk', d['outputs'].get('flags_filled_in_walk_rank0'), 'flag positions', d['outputs']['flag_positions_rank0'])"
done
