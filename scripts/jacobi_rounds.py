"""do the rounds of a cross-rank fixed-point walk converge early?  (GPU box; one GPU; measurement.)

The protocol measured: every rank walks its WHOLE shard at once on its best guess of the table it will be handed, the ranks exchange what their
walks added (their EFFECTS: keys created with stamps, distances by maximum, coverage increments, link flags -- the lattice writes of
src/ReadScanner.cpp:112-231), and rank r walks again on   base_r = T_0  join  effects of ranks 1 .. r-1 as of the round before,   until no rank's
table changes.  Rank r is exact after at most r rounds (induction over file order); the question is whether it is exact EARLIER, because only
then does the scheme beat the chain of N - 1 hand-overs (a round costs about what a hop costs: one walk of a shard + an exchange).

BASELINE config 2's reads as N file-order shards (6.25x coverage per shard at N = 8: config 4's per-rank coverage on 8 GPUs).  Every walk is the
library's exact ordered walk of one shard from a clean scan on an imported table; effects and joins are made on the host (numpy) -- this script
measures convergence, not speed.  Rank 0 always walks from the empty table and is exact; round 0 of the other ranks runs on `first_base`:
    T0    the complete table of rank 0 (they wait for it)
    hint  rank 0's table after a quarter of its shard (what scan_sharded shows them today)
    python scripts/jacobi_rounds.py [shards] [first_base]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from faucet_amd import _lib as L  # noqa: E402
from faucet_amd import api  # noqa: E402
from faucet_amd import synth_det as sd  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
FIRST = sys.argv[2] if len(sys.argv) > 2 else "T0"
dev = torch.device("cuda", 0)
k, E, S, G, R, Lr = 31, 100_000_000, 20_000_000, int(os.environ.get("JR_GENOME", 20_000_000)), int(os.environ.get("JR_READS", 10_000_000)), 100
reads = sd.make_reads(sd.make_genome(G, 2, dev), R, Lr, 0.01, 1000, dev)
tai, nh = api.load_filter_shape(E, S)
ctx = api.Context(k, tai, nh)
per = R // N
shards = [bench.device_batches(reads[r * per:(r + 1) * per], 625_000 if per >= 625_000 else per) for r in range(N)]
ctx.load_begin()
for s_ in shards:
    for b in s_:
        ctx.load_batch(b)
ctx.load_end()

ENTRY = np.dtype([("key", "<u8"), ("stamp", "<u8"), ("dist", "u1", 5), ("cov", "u1", 4), ("linked", "u1"), ("pad", "u1", 6)])
assert ENTRY.itemsize == L.TABLE_ENTRY_BYTES
STATS = [n for n, _ in L.ScanStats._fields_]


def walk(r, table, pieces_below, upto=None):
    """shard r from a clean scan on `table` (ENTRY array, any order); its pieces are numbered from `pieces_below` (creation stamps).
    Returns the table after it, sorted by key, and the scan's counters."""
    ctx.scan_begin()
    carried = {n: 0 for n in STATS}
    carried["reads_no_errors"] = pieces_below
    buf = torch.from_numpy(np.ascontiguousarray(table).view(np.uint8).reshape(-1).copy()).to(dev) if len(table) else torch.zeros(32, dtype=torch.uint8, device=dev)
    ctx.import_table(buf.data_ptr(), len(table), carried=carried)
    for b in shards[r][:upto]:
        ctx.scan_batch(b)
    st = ctx.scan_end()
    n = ctx.table_entries()
    out = torch.empty(max(n, 1) * L.TABLE_ENTRY_BYTES, dtype=torch.uint8, device=dev)
    got = ctx.export_table(out.data_ptr(), out.numel())
    t = out[:got * L.TABLE_ENTRY_BYTES].cpu().numpy().view(ENTRY).copy()
    t["pad"] = 0
    return t[np.argsort(t["key"], kind="stable")], st


def effects(s, base):
    """what the walk that turned `base` into `s` added (both sorted by key; base's keys are a subset of s's)"""
    pos = np.searchsorted(s["key"], base["key"])
    assert np.array_equal(s["key"][pos], base["key"])
    old = np.zeros(len(s), dtype=bool)
    old[pos] = True
    inc = s["cov"].astype(np.int16)
    inc[pos] -= base["cov"].astype(np.int16)
    raised = np.zeros(len(s), dtype=bool)
    raised[pos] = (s["dist"][pos] > base["dist"]).any(axis=1) | (s["linked"][pos] != base["linked"])
    keep = ~old | raised | (inc > 0).any(axis=1)
    return {"e": s[keep], "inc": inc[keep], "new": ~old[keep]}


def join(base, effs):
    """base joined with the effects of the ranks in `effs` (ascending rank): distances by maximum, coverage added up (saturating), link flags
    by OR, a key's record created by the earliest stamp"""
    keys = [base["key"]] + [e["e"]["key"] for e in effs]
    allk = np.unique(np.concatenate(keys))
    out = np.zeros(len(allk), dtype=ENTRY)
    out["key"] = allk
    out["stamp"] = np.iinfo(np.uint64).max
    cov = np.zeros((len(allk), 4), dtype=np.int32)
    p = np.searchsorted(allk, base["key"])
    out["stamp"][p], out["dist"][p], out["linked"][p] = base["stamp"], base["dist"], base["linked"]
    cov[p] = base["cov"]
    for e in effs:
        p = np.searchsorted(allk, e["e"]["key"])
        # (keys are unique within one rank's effects: plain fancy-index updates are safe)
        out["stamp"][p] = np.minimum(out["stamp"][p], np.where(e["new"], e["e"]["stamp"], np.iinfo(np.uint64).max))
        out["dist"][p] = np.maximum(out["dist"][p], e["e"]["dist"])
        out["linked"][p] |= e["e"]["linked"]
        cov[p] += e["inc"]
    out["cov"] = np.minimum(cov, 255).astype(np.uint8)
    # a key that only shows as "raised" -- created, a round ago, by a rank whose walk no longer creates it on its better base -- has no creator
    # left: it is not in the map (what was added to it is void; the rank that added it walks again on a base without it)
    return out[out["stamp"] != np.iinfo(np.uint64).max]


def differing(a, b, paths_only=False):
    """records of a and b (sorted by key) that are not equal (a key in one table only counts once).  paths_only: only what a walk's PATH can
    depend on -- which keys exist, and their distances"""
    both = np.intersect1d(a["key"], b["key"], assume_unique=True)
    pa, pb = np.searchsorted(a["key"], both), np.searchsorted(b["key"], both)
    neq = (a["dist"][pa] != b["dist"][pb]).any(axis=1)
    if not paths_only:
        neq |= (a["stamp"][pa] != b["stamp"][pb]) | (a["cov"][pa] != b["cov"][pb]).any(axis=1) | (a["linked"][pa] != b["linked"][pb])
    return int(neq.sum()) + (len(a) - len(both)) + (len(b) - len(both))


# ---- the sequential run: T_r = the table after shards 0 .. r; pieces of every shard (independent of the walk)
empty = np.zeros(0, dtype=ENTRY)
T, pieces, below = [], [], 0
t = empty
for r in range(N):
    t, st = walk(r, t, below)
    T.append(t)
    pieces.append(int(st["reads_no_errors"]) - below)
    below = int(st["reads_no_errors"])
    print(f"sequential: table after shard {r}: {len(t)} records", flush=True)
below = [sum(pieces[:r]) for r in range(N)]

if FIRST == "hint":
    quarter = max(1, len(shards[0]) // 4)
    first_base, _ = walk(0, empty, 0, upto=quarter)
    print(f"first base: rank 0's table after {quarter} of its {len(shards[0])} batches: {len(first_base)} records")
else:
    first_base = T[0]
print(f"\n{N} shards of {per} reads; round 0 of ranks >= 1 on {FIRST}.  Per round and rank: records of the BASE rank r walks on that differ from the sequential run's "
      f"T_(r-1) in what a path can depend on (keys, distances) / records of the table after its walk that differ from T_r in anything")
print("round | " + " | ".join(f"rank {r}" for r in range(1, N)) + " | ranks exact | every table as in the round before")
base = {r: first_base[np.argsort(first_base["key"], kind="stable")] for r in range(1, N)}
eff, prev_tables = {}, None
t_all = time.perf_counter()
for rnd in range(N):
    tables = {}
    for r in range(1, N):
        s, _ = walk(r, base[r], below[r])
        tables[r] = s
        eff[r] = effects(s, base[r])
    diffs = [differing(tables[r], T[r]) for r in range(1, N)]
    bdiffs = [differing(base[r], T[r - 1], paths_only=True) for r in range(1, N)]
    same = prev_tables is not None and all(len(tables[r]) == len(prev_tables[r]) and differing(tables[r], prev_tables[r]) == 0 for r in range(1, N))
    exact = 1 + sum(1 for d in diffs if d == 0)
    print(f"{rnd:5d} | " + " | ".join(f"{b:6d}/{d:6d}" for b, d in zip(bdiffs, diffs)) + f" | {exact} of {N} | {same}", flush=True)
    if same:
        break
    prev_tables = tables
    base = {r: join(T[0], [eff[q] for q in range(1, r)]) for r in range(1, N)}
print(f"({time.perf_counter() - t_all:.1f} s)")
