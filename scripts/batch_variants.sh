#!/bin/bash
# bench.py under different batchings (measurement aid): scripts/batch_variants.sh <outdir> "<bench args>" ...
out=gpurun_out/$1; shift
mkdir -p $out
for v in "$@"; do
  name=$(echo "$v" | tr ' =-' '___')
  timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu --no-ceilings --no-host-leg $v > $out/$name.json 2> $out/$name.err || echo "FAILED $v"
  python - "$out/$name.json" "$v" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = d["kernel_ms_per_step_rank0"]
    print(sys.argv[2], "| step", round(d["ms_per_step"], 2), "| walk_stage", k.get("walk_stage"), "flags", k.get("scan_flags"), "mark", k.get("load_mark"), "resolve", k.get("load_resolve"), "carry", k.get("carry_update"), "| windows", d["outputs"]["walk_windows_rank0"], flush=True)
except Exception as e:
    print(sys.argv[2], "no result:", e)
PY
done
