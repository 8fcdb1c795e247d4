"""Summary of a rocprofv3 --kernel-trace CSV of bench.py: for the LAST step (from the last k_pair_join... simply the last third of the
trace) per queue busy time, union busy time, idle gaps, and per-kernel overlapped durations."""
import csv
import glob
import os
import sys
from collections import defaultdict

rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].split("<")[0][:40], r.get("Queue_Id", "?")))
rows.sort()
# steps are delimited by k_pair_split (load_begin): take everything from the last one
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_pair_join")]
if len(starts) >= 1:
    rows = rows[starts[-1]:]
t0, t1 = rows[0][0], max(r[1] for r in rows)
print(f"last step: {len(rows)} dispatches, {(t1 - t0) / 1e6:.2f} ms from first start to last end")
# phases: load = until first scan kernel
first_scan = next((r[0] for r in rows if r[2].startswith("k_scan_valid") or r[2].startswith("k_scan_same")), t0)
print(f"load phase {(first_scan - t0) / 1e6:.2f} ms, scan phase {(t1 - first_scan) / 1e6:.2f} ms")
byq = defaultdict(list)
for r in rows:
    if r[0] >= first_scan:
        byq[r[3]].append(r)
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
allv = [(r[0], r[1]) for r in rows if r[0] >= first_scan]
print(f"scan phase: union of all kernels busy {union(allv) / 1e6:.2f} ms")
for q, rs in byq.items():
    iv = [(r[0], r[1]) for r in rs]
    names = defaultdict(lambda: [0, 0])
    for r in rs:
        names[r[2]][0] += 1; names[r[2]][1] += r[1] - r[0]
    top = sorted(names.items(), key=lambda kv: -kv[1][1])[:8]
    print(f"queue {q}: {len(rs)} dispatches, busy (union) {union(iv) / 1e6:.2f} ms, sum {sum(e - s for s, e in iv) / 1e6:.2f} ms, span {(max(e for s, e in iv) - min(s for s, e in iv)) / 1e6:.2f} ms")
    for n, (c, t) in top:
        print(f"      {n:40s} x{c:5d} {t / 1e6:8.2f} ms")
# gaps on the busiest queue pair: time where nothing runs
iv = sorted(allv); gaps = []; ce = iv[0][1]
for s, e in iv[1:]:
    if s > ce: gaps.append(s - ce)
    ce = max(ce, e)
print(f"idle gaps (no kernel on any queue) in scan phase: {len(gaps)} gaps, total {sum(gaps) / 1e6:.2f} ms, largest {max(gaps) / 1e6 if gaps else 0:.3f} ms")
