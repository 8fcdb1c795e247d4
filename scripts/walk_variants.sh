#!/bin/bash
# walk-stage variants on the main stream with per-kernel times (measurement aid): FGPU_WALK_WPS / FGPU_WALK_GENERIC / FGPU_MAX_SPAN_LOG2
out=gpurun_out/$1; shift
mkdir -p $out
for v in "$@"; do
  name=$(echo "$v" | tr ' =' '__')
  env $v FGPU_NO_OVERLAP=1 timeout -k 10 200 python bench.py --profile-walk --steps 3 --warmup 1 --no-cpu --no-ceilings > $out/$name.json 2> $out/$name.err || echo "FAILED $v"
  python - "$out/$name.json" "$v" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = d["kernel_ms_per_step_rank0"]
    print(sys.argv[2], "| step", round(d["ms_per_step"], 1), "| walk_stage", k.get("walk_stage"), "walk", k.get("walk"), "lookup", k.get("walk_lookup"), "link", k.get("walk_link"),
          "clean", k.get("walk_clean"), "cluster", k.get("walk_cluster"), "| windows", d["outputs"]["walk_windows_rank0"], "junctions", d["outputs"]["junctions"], flush=True)
except Exception as e:
    print(sys.argv[2], "no result:", e)
PY
done
