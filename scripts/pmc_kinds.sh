#!/bin/bash
# Runs ON THE GPU BOX: counters of k_load_mark over several contexts of ONE process (scripts/kinds_probe.py contexts N: the kernel's
# speed changes from context to context and stays within one), with the kernel trace in the same run so that every dispatch has its duration beside its
# counters.  One counter set per run (no other trace domain).      gpurun -- 'bash scripts/pmc_kinds.sh 6'
n=${1:-6}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_kinds
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_ATOMIC_sum TCC_EA0_ATOMIC_LEVEL_sum" \
           "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum"; do
  i=$((i + 1))
  rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "k_load_mark" --output-format csv -d "$out/s$i" -o run -- python3 "$root/scripts/kinds_probe.py" --child contexts $n > "$out/s$i.log" 2>&1
  grep "load_mark per pass" "$out/s$i.log" | sed "s/^/[set $i] /"
done
find "$out" -name "*.db" -delete
python3 - "$out" $n <<'PY'
import csv, glob, os, sys
out, n = sys.argv[1], int(sys.argv[2])
for d in sorted(glob.glob(os.path.join(out, "s?"))):
    trace = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    cnt = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not trace or not cnt:
        print(d, "no output"); continue
    dur = {}
    for row in csv.DictReader(open(trace[0])):
        if "k_load_mark" in row["Kernel_Name"]:
            dur[row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
    per = {}
    for row in csv.DictReader(open(cnt[0])):
        per.setdefault(row["Dispatch_Id"], {})[row["Counter_Name"]] = float(row["Counter_Value"])
    ids = sorted(per, key=int)
    per_ctx = len(ids) // n if n else len(ids)
    names = sorted({c for v in per.values() for c in v})
    print(os.path.basename(d), "dispatches", len(ids), "per context", per_ctx)
    for c in range(n):
        chunk = ids[c * per_ctx:(c + 1) * per_ctx]
        chunk = chunk[2 * (per_ctx // 3):]          # the context's third pass
        ms = sum(dur.get(i, 0.0) for i in chunk)
        line = f"  context {c}: load_mark {ms:8.2f} ms (third pass)"
        for nm in names:
            line += f"  {nm} {sum(per[i].get(nm, 0.0) for i in chunk):.4g}"
        print(line)
PY
