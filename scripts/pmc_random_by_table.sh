#!/bin/bash
# Runs ON THE GPU BOX: why do first-set-time atomics lose a third of their rate on a 32 GiB table?
# Address-translation (UTCL1 / UTCL2) and memory-side (TCC_EA0) counters of k_diag_random on 2 GiB and 32 GiB tables, loads and atomicMin,
# then the same counters for k_load_mark on config 2 and on config 4's per-GPU shape.  One --pmc set per run, never with a trace domain.
#   gpurun --timeout 1100 -- 'bash scripts/pmc_random_by_table.sh'
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_tables
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
sets=(
 "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum"
 "TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_LFIFO_FULL_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_SERIALIZATION_STALL_sum"
 "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum TCC_REQ_sum"
 "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_HIT_sum"
 "TCC_EA0_ATOMIC_LEVEL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum GRBM_GUI_ACTIVE"
 "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum"
)
i=0
for s in "${sets[@]}"; do
  rocprofv3 --pmc $s --kernel-include-regex "k_diag_random" --output-format csv -d "$out/diag$i" -o run -- python3 "$root/scripts/pmc_random_by_table.py" > "$out/diag$i.txt" 2> "$out/diag$i.err" || echo "set $i failed"
  i=$((i+1))
done
# the product kernel: config 2 (2 GiB of first-set times) and config 4's per-GPU shape (25 M reads, 2^33-bit filters: 32 GiB of times)
i=0
for s in "${sets[@]:0:4}"; do
  rocprofv3 --pmc $s --kernel-include-regex "k_load_mark" --output-format csv -d "$out/c2_$i" -o run -- python3 "$root/bench.py" --steps 1 --warmup 0 --no-cpu --no-ceilings --no-host-leg > "$out/c2_$i.json" 2> "$out/c2_$i.err" || echo "c2 set $i failed"
  rocprofv3 --pmc $s --kernel-include-regex "k_load_mark" --output-format csv -d "$out/c4_$i" -o run -- python3 "$root/bench.py" --reads 25000000 --genome 400000000 --estimated-kmers 1000000000 --singletons 200000000 --batch-reads 2500000 --steps 1 --warmup 0 --no-cpu --no-ceilings --no-host-leg > "$out/c4_$i.json" 2> "$out/c4_$i.err" || echo "c4 set $i failed"
  i=$((i+1))
done
find "$out" \( -name "*.db" \) -delete
python3 - "$out" <<'PY'
import csv, glob, os, sys
root = sys.argv[1]
for d in sorted(glob.glob(os.path.join(root, "*"))):
    if not os.path.isdir(d):
        continue
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    if not rows:
        print(os.path.basename(d), "no counter rows")
        continue
    by = {}
    for r in rows:
        by.setdefault((int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0][-28:]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    tag = os.path.basename(d)
    if tag.startswith("diag"):
        for (disp, k), c in sorted(by.items()):
            print(tag, disp, k, " ".join(f"{n}={v:.4g}" for n, v in sorted(c.items())))
    else:   # many launches: sums
        acc, n = {}, 0
        for (_, k), c in by.items():
            n += 1
            for name, v in c.items():
                acc[name] = acc.get(name, 0.0) + v
        print(tag, f"launches={n}", " ".join(f"{name}={v:.4g}" for name, v in sorted(acc.items())))
PY
