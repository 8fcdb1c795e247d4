#!/bin/bash
# The library's measurement knobs, tuned on config 2 in rounds 1-2, looked at again on a large-filter configuration (config 5's shape: the
# quickest of them).  One knob at a time against the defaults; per-step time and the kernels a knob can move.  GPU box.
cd ${GRAFT_REPO_ROOT:-.}
args="--reads 50000000 --read-len 150 --genome 150000000 --estimated-kmers 2000000000 --singletons 1000000000 --err 0.05 --batch-reads 2000000 --steps 1 --warmup 1 --no-cpu --no-ceilings --no-host-leg"
one() {
  env "$@" timeout -k 10 200 python bench.py $args > /tmp/ks.json 2>/dev/null
  python - "$*" <<'PY'
import json, sys
d = json.loads(open("/tmp/ks.json").read().strip().splitlines()[-1]); k = d["kernel_ms_per_step_rank0"]
print("%-28s %6.0f ms per step; load_mark %4.0f walk_stage %4.0f scan_flags %3.0f scan_valid %3.0f need_lookup %3.0f load_resolve %3.0f carry %3.0f; junctions %d" % (sys.argv[1] or "defaults", d["ms_per_step"], k["load_mark"], k["walk_stage"], k["scan_flags"], k["scan_valid"], k["need_lookup"], k["load_resolve"], k["carry_update"], d["outputs"]["junctions"]), flush=True)
PY
}
one X=1
one FGPU_MAX_SPAN_LOG2=25
one FGPU_MAX_SPAN_LOG2=27
one FGPU_NEED_TIGHT=0
one FGPU_NEED_TIGHT=1
one FGPU_SCAN_BUFFERS=3
one FGPU_SWEEP_RATIO=1/2
one FGPU_SWEEP_RATIO=2/1
one FGPU_RESOLVE_SM=0
one X=2
