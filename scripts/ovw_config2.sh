for v in "FGPU_X=0" "FGPU_WALK_KO=64 FGPU_WALK_KO_ALWAYS=1" "FGPU_WALK_KO=32 FGPU_WALK_KO_ALWAYS=1" "FGPU_WALK_KO=16 FGPU_WALK_KO_ALWAYS=1" "FGPU_WALK_KO=8 FGPU_WALK_KO_ALWAYS=1" "FGPU_WALK_KO=32 FGPU_WALK_KO_ALWAYS=1 FGPU_WALK_KO_WEIGHT=128"; do
  env $v python3 bench.py --steps 10 --warmup 3 --no-cpu --no-ceilings --no-host-leg --no-full-size > /tmp/b.json 2> /tmp/b.err
  python3 - "$v" <<'PY'
import json, sys
d = json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
k = d["kernel_ms_per_step_rank0"]
print(f"config2 [{sys.argv[1]}] {d['ms_per_step']:.2f} ms/step  walk_stage {k.get('walk_stage')}  ovw pieces {d['outputs']['walk_key_ordered_pieces_rank0']}  max cluster {d['outputs']['walk_max_cluster_rank0']} junctions {d['outputs']['junctions']}")
PY
done
