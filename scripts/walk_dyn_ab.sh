#!/bin/bash
# Runs ON THE GPU BOX: k_walk_dyn (clusters handed out dynamically, one piece per lane and round) against k_walk (lane i walks the cluster led by piece i):
# config 2's step and config 4's per-GPU shape.  Measurement aid.
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
c4="--reads 25000000 --genome 400000000 --estimated-kmers 1000000000 --singletons 200000000 --batch-reads 2500000"
for v in "FGPU_WALK_STATIC=1" "FGPU_X=0" "FGPU_WALK_DYN_GRID=1024" "FGPU_WALK_DYN_GRID=4096"; do
  for shape in "" "$c4"; do
    env $v FGPU_PROFILE_WALK=1 python3 bench.py --steps 6 --warmup 2 --no-cpu --no-ceilings --no-host-leg --no-full-size $shape > /tmp/b.json 2> /tmp/b.err
    python3 - "$v" "$shape" <<'PY'
import json, sys
d = json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
k = d["kernel_ms_per_step_rank0"]
print(f"[{sys.argv[1]}] {'config4 per-GPU shape' if sys.argv[2] else 'config2'}: {d['ms_per_step']:.2f} ms/step  walk_stage {k.get('walk_stage')}  walk {k.get('walk')}  junctions {d['outputs']['junctions']}", flush=True)
PY
  done
done
