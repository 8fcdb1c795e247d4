#!/bin/bash
# Runs ON THE GPU BOX: HBM fetch bytes and L2 hit/miss of the direct vs binned probe kernels (fgpu_diag_binned_probes), one table size
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_binned
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
cat > /tmp/ab_one.py <<PY
import sys
sys.path.insert(0, "$root")
from faucet_amd import api
ctx = api.Context(31, 1 << 29, 3)
for table in (64 << 20, 512 << 20):
    print(table >> 20, ctx.diag_binned_probes(table, 1 << 28, 4 << 20, 1))
PY
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_diag_" --output-format csv -d "$out/fetch" -o run -- python3 /tmp/ab_one.py > "$out/fetch.log" 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-include-regex "k_diag_" --output-format csv -d "$out/l2" -o run -- python3 /tmp/ab_one.py > "$out/l2.log" 2>&1
python3 - "$out" <<'PY'
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("(anonymous namespace)::", ""), r["Counter_Name"], float(r["Counter_Value"])))
rows.sort()
for d, k, c, v in rows:
    print(f"dispatch {d:3d} {k:24s} {c:14s} {v:.4g}")
PY
