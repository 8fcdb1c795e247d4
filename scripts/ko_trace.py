"""Where the key-ordered walk's time goes, piece by piece (GPU box; library built with FGPU_EXTRA_CXXFLAGS=-DFGPU_KO_TRACE):
    python scripts/ko_trace.py [N_PAIRS] [repeats] [repeat_len]
Runs the repeat-rich paired-end set of scripts/pe_profile.py once and reads back one record per piece the key-ordered walk walked: start, end,
time spent waiting for turns, lk positions.  Prints per-piece work (duration - waits), how many pieces run / wait at a time, and the share of
the walk during which NO piece is doing work (everybody waits: a hand-over is in flight)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from faucet_amd import api, synth  # noqa: E402

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 2_500_000
repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rlen = int(sys.argv[3]) if len(sys.argv) > 3 else 400
g = synth.make_genome(6_000_000, 31, repeats=repeats, repeat_len=rlen)
r = synth.make_pairs(g, n_pairs, 100, 300, 30, 0.01, 32)
reads = torch.from_numpy(np.ascontiguousarray(r)).to("cuda:0")
n = r.shape[0]
tai, nh = api.load_filter_shape(8 * n, 2 * n)
batches = bench.device_batches(reads, 600_000)
ctx = api.Context(31, tai, nh, profile=True, key_order_from_start=True)
ctx.load_begin()
for b in batches:
    ctx.load_batch(b)
ctx.load_end()
st = api.ReadScanner(ctx).scanReads(batches)
t = ctx.diag_ko_trace()
print({k: st[k] for k in ("walk_windows", "walk_max_cluster", "walk_parallel", "n_junctions")})
if not len(t):
    raise SystemExit("no trace: build with FGPU_EXTRA_CXXFLAGS=-DFGPU_KO_TRACE")
start, end = t[:, 1].astype(np.int64), t[:, 2].astype(np.int64)
wait = (t[:, 3] & np.uint64((1 << 48) - 1)).astype(np.int64)
nlk = (t[:, 3] >> np.uint64(48)).astype(np.int64)
dur = end - start
work = dur - wait
us = 0.01
print(f"pieces {len(t)}; per piece: duration {dur.mean() * us:.1f} us, waiting {wait.mean() * us:.1f} us, work {work.mean() * us:.1f} us "
      f"(median {np.median(work) * us:.1f}, p90 {np.percentile(work, 90) * us:.1f}); lk positions {nlk.mean():.1f}; work per lk position {work.sum() / max(nlk.sum(), 1) * us:.2f} us")
for lo, hi in ((0, 4), (4, 16), (16, 48), (48, 600)):
    m = (nlk >= lo) & (nlk < hi)
    if m.any():
        print(f"  pieces with {lo:3d}-{hi:3d} lk positions: {int(m.sum()):6d}, work {work[m].mean() * us:7.1f} us, waiting {wait[m].mean() * us:8.1f} us")
# timeline: sweep over start/end events; busy = pieces in flight, working = in flight minus waiting (approximated: waits are contiguous per piece? no --
# so only the totals are exact): total span of the walks, sum of work, sum of waits
order = np.argsort(start)
gaps = np.diff(np.sort(start))
span = int(end.max() - start.min())
# windows show as gaps in the start times (a new launch): split at gaps > 200 us
cuts = np.nonzero(gaps > 20000)[0]
bounds = np.concatenate([[0], cuts + 1, [len(t)]])
ss, ee, ww = np.sort(start), end[order], work[order]
tot_span = 0
for a, b in zip(bounds[:-1], bounds[1:]):
    tot_span += int(ee[a:b].max() - ss[a:b].min())
print(f"launches {len(bounds) - 1}: time covered by piece walks {tot_span * us / 1000:.1f} ms, sum of work {work.sum() * us / 1000:.1f} ms, sum of waits {wait.sum() * us / 1000:.1f} ms "
      f"-> pieces doing work at a time {work.sum() / max(tot_span, 1):.2f}, pieces in flight at a time {dur.sum() / max(tot_span, 1):.1f}")
np.save("gpurun_out/ko_trace.npy", t)

# ---- per-step stamps of one piece in 16: where inside a position the time goes
S = ctx.diag_ko_stamps()
if len(S):
    names = {1: "account: enter", 2: "account: turn held", 3: "lookup done", 4: "stop chosen", 5: "visit done", 6: "passed on"}
    gaps = {}       # (from code, to code) -> list of tick differences
    per_piece = []
    for row in S:
        nst = int(row[0] & np.uint64(0xFFFF))
        if nst < 2:
            continue
        w = row[1:1 + min(nst, 1020)]
        code = (w >> np.uint64(56)).astype(np.int64)
        tk = (w & np.uint64(0xFFFFFFFFFF)).astype(np.int64)
        d = np.diff(tk)
        for a, b, dt in zip(code[:-1], code[1:], d):
            gaps.setdefault((int(a), int(b)), []).append(int(dt))
        per_piece.append((nst, int(tk[-1])))
    print(f"stamped pieces {len(per_piece)}; stamps per piece {np.mean([p[0] for p in per_piece]):.0f}; last stamp at {np.mean([p[1] for p in per_piece]) * us:.1f} us")
    tot = sum(sum(v) for v in gaps.values())
    for (a, b), v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:14]:
        v = np.array(v)
        print(f"  {names.get(a, a):22s} -> {names.get(b, b):22s}: {len(v):7d} x  mean {v.mean() * us:7.2f} us  median {np.median(v) * us:7.2f}  p90 {np.percentile(v, 90) * us:7.2f}   {100 * v.sum() / tot:5.1f} % of the stamped time")
    np.save("gpurun_out/ko_stamps.npy", S)
