#!/bin/bash
# Config 2 with clusters of at least T pieces on the key-ordered walk, T = off / 64 / 32 / 16 / 8 (GPU box; measurement aid).
cd ${GRAFT_REPO_ROOT:-.}
for t in off 64 32 16 8; do
  if [ $t = off ]; then env="FGPU_WALK_KO=0"; else env="FGPU_WALK_KO=$t FGPU_WALK_KO_ALWAYS=1"; fi
  env $env timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu --no-ceilings --no-host-leg > /tmp/ko_$t.json 2>/dev/null
  python - <<PY
import json
d = json.loads(open("/tmp/ko_$t.json").read().strip().splitlines()[-1])
k = d["kernel_ms_per_step_rank0"]
print("threshold $t: %.1f ms per step, walk_stage %.1f ms, key-ordered pieces %d, junctions %d" % (d["ms_per_step"], k["walk_stage"], d["outputs"]["walk_key_ordered_pieces_rank0"], d["outputs"]["junctions"]), flush=True)
PY
done
