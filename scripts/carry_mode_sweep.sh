cd ${GRAFT_REPO_ROOT:-.}
run() { # name, extra bench args...
  name=$1; shift
  for mode in set sweep; do
    for ratio in 1/1 1/2; do
      if [ $mode = set ] && [ $ratio != 1/1 ]; then continue; fi
      FGPU_CARRY_MODE=$mode FGPU_SWEEP_RATIO=$ratio timeout -k 10 300 python bench.py "$@" --steps 1 --warmup 1 --no-cpu --no-ceilings --no-host-leg > /tmp/cs.json 2>/dev/null
      python - <<PY
import json
d = json.loads(open("/tmp/cs.json").read().strip().splitlines()[-1]); k = d["kernel_ms_per_step_rank0"]
print("$name | carry $mode, sweep ratio $ratio: %.0f ms per step; load_mark %.0f, carry_update %.0f, load_resolve %.0f; junctions %d, to_bloo2 %d" % (d["ms_per_step"], k["load_mark"], k["carry_update"], k["load_resolve"], d["outputs"]["junctions"], d["outputs"]["to_bloo2_rank0"]), flush=True)
PY
    done
  done
}
run "config 5 (50 M x 150, 2^33 bits, 2 hashes)" --reads 50000000 --read-len 150 --genome 150000000 --estimated-kmers 2000000000 --singletons 1000000000 --err 0.05 --batch-reads 2000000
run "config 4's per-GPU share (25 M reads, 2^33 bits)" --reads 25000000 --genome 400000000 --estimated-kmers 1000000000 --singletons 200000000 --batch-reads 2500000
run "one rank's reads of the 8-rank bench (10 M reads, 2^32 bits)" --reads 10000000 --genome 160000000 --estimated-kmers 800000000 --singletons 160000000
run "one rank's reads of the 4-rank bench (10 M reads, 2^31 bits)" --reads 10000000 --genome 80000000 --estimated-kmers 400000000 --singletons 80000000
