"""Where one hop of the strong-scaling chain spends its wall time: a rocprofv3 --kernel-trace CSV of scripts/hop_kernels.py, cut to the LAST
hop (from the import of the handed-over table to the start of the export behind the walk): span, time with at least one kernel running,
idle gaps by length, and the kernels' own sums.
    python scripts/hop_timeline.py <dir with *kernel_trace.csv>"""
import csv
import glob
import os
import sys
from collections import defaultdict

rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].split("<")[0][:40]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "?")))
rows.sort()
exports = [i for i, r in enumerate(rows) if r[2] == "k_export"]
if not exports:
    sys.exit("no k_export in the trace")
end_i = exports[-1]
probes = [i for i, r in enumerate(rows[:end_i]) if r[2] == "k_import_probe"]
begin_i = probes[-1]
hop = rows[begin_i:end_i]
t0, t1 = hop[0][0], rows[end_i][0]
print(f"hop: {len(hop)} dispatches, {(t1 - t0) / 1e6:.2f} ms from the import's first kernel to the export's start")
iv = sorted((s, e) for s, e, _, _ in hop)
busy, gaps, ce = 0, [], iv[0][0]
cs = iv[0][0]
ce = iv[0][1]
for s, e in iv[1:]:
    if s > ce:
        busy += ce - cs
        gaps.append((s - ce, ce))
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print(f"at least one kernel running: {busy / 1e6:.2f} ms; idle: {sum(g for g, _ in gaps) / 1e6:.2f} ms in {len(gaps)} gaps")
for lo, hi in ((0, 5e3), (5e3, 2e4), (2e4, 1e5), (1e5, 1e6), (1e6, 1e12)):
    sel = [g for g, _ in gaps if lo <= g < hi]
    print(f"    gaps of {lo / 1e3:.0f}-{hi / 1e3:.0f} us: {len(sel):5d}, {sum(sel) / 1e6:7.2f} ms")
names = defaultdict(lambda: [0, 0])
for s, e, n, _ in hop:
    names[n][0] += 1
    names[n][1] += e - s
for n, (c, t) in sorted(names.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"    {n:40s} x{c:5d} {t / 1e6:8.2f} ms")
# the largest gaps: which kernels stand on either side
big = sorted(gaps, reverse=True)[:12]
ends = {e: n for s, e, n, _ in hop}
starts = sorted((s, n) for s, e, n, _ in hop)
for g, at in big:
    nxt = next((n for s, n in starts if s >= at + g), "?")
    print(f"    gap {g / 1e3:8.1f} us after {ends.get(at, '?'):28s} before {nxt}")

# HOP_DUMP=n: the first n dispatches of the hop one by one (start since the hop's beginning, duration, queue, kernel)
n_dump = int(os.environ.get("HOP_DUMP", "0"))
for st, en, name, q in hop[:n_dump]:
    print(f"    {(st - t0) / 1e3:9.1f} us  +{(en - st) / 1e3:8.1f} us  q{q}  {name}")
