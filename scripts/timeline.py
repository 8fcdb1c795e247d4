"""Timeline of one step from a rocprofv3 kernel trace (CSV): per HIP stream (queue) busy time, how much of it overlaps with the other streams, and
the gaps of the whole device -- where pass 2's wall time goes beyond its kernels.
  python scripts/timeline.py <kernel_trace.csv> [step_index_from_the_end=1]"""
import csv
import re
import sys
from collections import defaultdict

path = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), (re.search(r"\b(k_\w+)", r["Kernel_Name"]) or re.search(r"(\w+)", r["Kernel_Name"])).group(1), r.get("Queue_Id", "?"), r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
# steps: every step starts with the first k_pack after a k_pair_split (load_end of the previous pass 1) ... simpler: split at k_pair_join (load_begin)
starts = [i for i, r in enumerate(rows) if r[2] == "k_pair_join"]
if len(starts) < back + 1:
    sys.exit("not enough steps in the trace")
lo = starts[-back - 1]
hi = starts[-back] if back else len(rows)       # 0 = the last step of the trace, to its end
step = rows[lo:hi]
t0 = step[0][0]
split = next(i for i, r in enumerate(step) if r[2] == "k_pair_split")        # load_end: pass 1 | pass 2
for name, part in (("pass 1", step[:split + 1]), ("pass 2", step[split + 1:])):
    if not part:
        continue
    a, b = part[0][0], max(r[1] for r in part)
    print(f"== {name}: {(b - a) / 1e6:.2f} ms wall, {len(part)} dispatches, kernel time {sum(r[1] - r[0] for r in part) / 1e6:.2f} ms")
    by = defaultdict(list)
    for r in part:
        by[r[4]].append(r)
    # device-level: union of busy intervals, and time with >= 2 kernels in flight
    ev = []
    for r in part:
        ev.append((r[0], 1))
        ev.append((r[1], -1))
    ev.sort()
    depth, last, busy, multi, idle_gaps = 0, a, 0, 0, []
    for t, d in ev:
        if depth >= 1:
            busy += t - last
        if depth >= 2:
            multi += t - last
        if depth == 0 and t - last > 20000:
            idle_gaps.append((last - a, t - last))
        depth += d
        last = t
    print(f"   device busy {busy / 1e6:.2f} ms, two or more kernels in flight {multi / 1e6:.2f} ms, idle {((b - a) - busy) / 1e6:.2f} ms; idle gaps > 20 us: {len(idle_gaps)}, "
          f"largest {sorted(idle_gaps, key=lambda g: -g[1])[:5]}")
    for q, rs in sorted(by.items(), key=lambda kv: -sum(r[1] - r[0] for r in kv[1])):
        kt = defaultdict(float)
        for r in rs:
            kt[r[2]] += (r[1] - r[0]) / 1e6
        span = (max(r[1] for r in rs) - min(r[0] for r in rs)) / 1e6
        top = ", ".join(f"{n} {ms:.1f}" for n, ms in sorted(kt.items(), key=lambda kv: -kv[1])[:7])
        print(f"   stream {q}: {len(rs)} dispatches, busy {sum(r[1] - r[0] for r in rs) / 1e6:.2f} ms over a span of {span:.2f} ms (first at {(min(r[0] for r in rs) - a) / 1e6:.2f} ms) | {top}")
    if name == "pass 2" and len(sys.argv) > 3:
        for r in part:
            print(f"      {(r[0] - a) / 1e6:8.3f} {(r[1] - r[0]) / 1e6:7.3f} {r[4]:>4} {r[2]}")
