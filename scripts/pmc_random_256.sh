#!/bin/bash
# Runs ON THE GPU BOX: the memory-side counters of scripts/pmc_random_by_table.sh again with a 256 MiB table beside the 2 GiB and 32 GiB ones
# (k_diag_random only): is the atomics' rate lost where the table outgrows the Infinity Cache?
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_tables256
mkdir -p "$out"
export TMPDIR=/tmp FGPU_PMC_WITH_256MIB=1
cd /tmp
sets=(
 "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum GRBM_UTCL2_BUSY"
 "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum TCC_REQ_sum"
 "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_HIT_sum"
 "TCC_EA0_ATOMIC_LEVEL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum GRBM_GUI_ACTIVE"
 "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_ATOMIC_DRAM_sum TCC_EA0_WRREQ_WRITE_DRAM_sum"
)
i=0
for s in "${sets[@]}"; do
  rocprofv3 --pmc $s --kernel-include-regex "k_diag_random" --output-format csv -d "$out/diag$i" -o run -- python3 "$root/scripts/pmc_random_by_table.py" > "$out/diag$i.txt" 2> "$out/diag$i.err" || echo "set $i failed"
  i=$((i+1))
done
find "$out" \( -name "*.db" \) -delete
cat "$out/diag0.txt"
python3 - "$out" <<'PY'
import csv, glob, os, sys
for d in sorted(glob.glob(os.path.join(sys.argv[1], "diag*"))):
    if not os.path.isdir(d):
        continue
    by = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            by.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    for disp, c in sorted(by.items()):
        print(os.path.basename(d), disp, " ".join(f"{n}={v:.4g}" for n, v in sorted(c.items())))
PY
