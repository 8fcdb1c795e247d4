#!/bin/bash
# overlapped bench under different environment settings (measurement aid): scripts/env_variants.sh <outdir> "VAR=val VAR2=val" ...
out=gpurun_out/$1; shift
mkdir -p $out
for v in "$@"; do
  name=$(echo "$v" | tr ' =/' '___')
  env $v timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu --no-ceilings > $out/$name.json 2> $out/$name.err || echo "FAILED $v"
  python - "$out/$name.json" "$v" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = d["kernel_ms_per_step_rank0"]
    print(sys.argv[2], "| step", round(d["ms_per_step"], 2), "| walk_stage", k.get("walk_stage"), "flags", k.get("scan_flags"), "valid", k.get("scan_valid"), "need", k.get("need_lookup"),
          "mark", k.get("load_mark"), "resolve", k.get("load_resolve"), "| windows", d["outputs"]["walk_windows_rank0"], "junctions", d["outputs"]["junctions"], flush=True)
except Exception as e:
    print(sys.argv[2], "no result:", e)
PY
done
