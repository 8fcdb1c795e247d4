"""Projected step of `bench.py --gpus N --scaling strong` (BASELINE config 4 cut into N file-order shards) from per-rank stage times measured on ONE
MI355X: the N ranks' work is executed in turn by one process with the very backend calls of faucet_amd/sharded.py (GpuShard), every stage
bracketed by a device synchronisation; transfers are priced, not measured (LINK GB/s per xGMI link and direction, every rank on its own link to
every other).  The projection is the critical path of sharded.load_sharded_presence + sharded.scan_sharded:
    pass 1 = max over ranks (own load + fix-up, or presence pass + load on the carried-in prefix: what sharded.load_sharded takes) + the two
             slice-wise exchanges
    pass 2 = max(rank 0's streaming scan, the others' pure stage) + sum over ranks > 0 of (table transfer + import + walk + export)
  python scripts/project_strong.py [N ...]       (default 2 4 8; FIXTURE=config4)"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from faucet_amd import _lib as L  # noqa: E402
from faucet_amd import api, sharded  # noqa: E402
from faucet_amd import synth_det as sd  # noqa: E402

LINK = float(os.environ.get("LINK_GBPS", "50"))
LATE = sharded.LATE_HINT          # FAUCET_LATE_HINT=0: as until round 5
WARM = int(os.environ.get("WARM", "2"))        # every stage runs WARM times, the last one is the one reported (a rank of a bench run is warm: the timed steps follow warm-up steps)
Ns = [int(a) for a in sys.argv[1:]] or [2, 4, 8]
fx = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "fullsize.json")))[os.environ.get("FIXTURE", "config4")]
c = fx["params"]
dev = torch.device("cuda", 0)
tai, nh = api.load_filter_shape(c["E"], c["S"])
genome = sd.make_genome(c["genome"], c["genome_seed"], dev)
kmers = c["reads"] * (c["read_len"] - c["k"] + 1)


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return out, 1e3 * (time.perf_counter() - t0)


for N in Ns:
    per = c["reads"] // N
    def shard_reads(r):
        return sd.make_reads(genome, per, c["read_len"], c["err"], c["read_seed"], dev, first_row=r * per)
    ctx = api.Context(c["k"], tai, nh)
    b = sharded.GpuShard(ctx, dev, stream_ordered=False)
    # ---- pass 1 with the protocol sharded.load_sharded takes for this shape (FAUCET_SHARD_PROTOCOL overrides it as it does there)
    probe = bench.device_batches(shard_reads(0), bench.batch_bounds(per, 2_500_000, 2))
    fixup = os.environ.get("FAUCET_SHARD_PROTOCOL", "auto") != "presence" and b.fixup_possible(probe)
    del probe
    t_first, t_second, prefixes = [], [], []     # presence pass / own load;  load on the prefix / fix-up
    running = acc2 = None
    if fixup:
        for r in range(N):
            reads = shard_reads(r)
            batches = bench.device_batches(reads, bench.batch_bounds(per, 2_500_000, 2))
            for _ in range(WARM):
                b.clear_filters()
                _, ms = timed(lambda: b.load(batches, keep_carry=False, shard_times=True))
            t_first.append(ms)
            if running is None:
                running = torch.zeros_like(b.bloom_tensor(L.BLOO1))
                acc2 = torch.zeros_like(running)
            prefixes.append(running.clone())
            ms = 0.0
            if r > 0:
                _, ms = timed(lambda: b.load_fixup(prefixes[r]))
            t_second.append(ms)
            running |= b.bloom_tensor(L.BLOO1)
            acc2 |= b.bloom_tensor(L.BLOO2)
            del reads, batches
    else:
        pres = []
        for r in range(N):
            reads = shard_reads(r)
            batches = bench.device_batches(reads, bench.batch_bounds(per, 2_500_000, 2))
            for _ in range(WARM):
                b.clear_filters()
                _, ms = timed(lambda: [b.presence(x) for x in batches])
                ctx.synchronize()
            t_first.append(ms)
            pres.append(b.bloom_tensor(L.BLOO1).clone())
            del reads, batches
        running = torch.zeros_like(pres[0])
        acc2 = torch.zeros_like(pres[0])
        for r in range(N):
            reads = shard_reads(r)
            batches = bench.device_batches(reads, bench.batch_bounds(per, 2_500_000, 2))
            for _ in range(WARM):
                b.clear_filters()
                b.bloom_tensor(L.BLOO1).copy_(running)
                ctx.synchronize()
                _, ms = timed(lambda: b.load(batches, keep_carry=True))
            t_second.append(ms)
            prefixes.append(running.clone())
            running |= pres[r]
            acc2 |= b.bloom_tensor(L.BLOO2)
            del reads, batches
        pres = None
    # the reduced filters are the whole run's: checked against the oracle's digests (tests/golden/fullsize.json) before pass 2 uses them
    import hashlib
    d1, d2 = (hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest() for t in (running, acc2))
    filters_ok = d1 == fx.get("bloo1_sha256") and d2 == fx.get("bloo2_sha256")
    # ---- pass 2: rank 0 streams; rank r prepares on the hint (rank 0's table after a quarter of its reads), then imports, walks, exports
    t_pure, t_hop, recs, t_late, n_merged = [], [], [], [], []
    older = None
    table, n_table, stats, hint = None, 0, None, None
    t_scan0 = 0.0
    for r in range(N):
        reads = shard_reads(r)
        batches = bench.device_batches(reads, bench.batch_bounds(per, 2_500_000, 2))
        # a rank scans the reads it has just loaded: the planes the load pass keeps ("this occurrence went to bloo2") answer part of the scan's
        # validity probes, so the rank's own load comes first here too (untimed), then the reduced bloo2 takes the place of the local one
        b.clear_filters()
        if fixup:
            b.load(batches, keep_carry=False, shard_times=True)
            if r > 0:
                b.load_fixup(prefixes[r])
        else:
            b.bloom_tensor(L.BLOO1).copy_(prefixes[r])
            ctx.synchronize()
            b.load(batches, keep_carry=True)
        b.bloom_tensor(L.BLOO2).copy_(acc2)
        ctx.synchronize()
        prev_stats = stats
        for rep in range(WARM):
            b.scan_begin()
            if r == 0:
                total, done, marks, marks_late = per, 0, [], []
                for x in batches:
                    done += x.n_reads
                    marks.append(done >= sharded.HINT_AFTER * total)
                    marks_late.append(done >= sharded.LATE_AFTER * total)
                hi, hi_late = marks.index(True), marks_late.index(True)
                hint = None
                late0 = None
                def show(i):
                    global hint, late0
                    if hint is None and i >= hi:
                        n, buf = b.export_table(tag="hint")
                        hint = (buf[:max(n, 1) * L.TABLE_ENTRY_BYTES].clone(), n)
                    if late0 is None and i >= hi_late and LATE and N > 1:      # (round 6: the second rank's fresher preview; its export is inside rank 0's timed scan)
                        n, buf = b.export_table(tag="late0")
                        late0 = (buf[:max(n, 1) * L.TABLE_ENTRY_BYTES].clone(), n)
                stats, t_scan0 = timed(lambda: b.scan_stream(batches, after_batch=show))
                older = late0
            else:
                b.import_hint(hint[0], hint[1])
                _, ms_pure = timed(lambda: [b.scan_prepare(x) for x in batches])
                ms_late = 0.0
                if r >= 1 and LATE and older is not None:      # the table the rank below was handed (rank 1: rank 0's table after LATE_AFTER of its reads), passed on as a fresher preview: planes made again off the chain
                    _, ms_late = timed(lambda: (b.import_hint(older[0], older[1]), b.refresh_prepared()))
                carried = {n: int(prev_stats[n]) for n in sharded._STAT_NAMES}
                (stats), ms_walk = timed(lambda: b.walk_shard(batches, table, n_table, carried))
                merged = ctx.diag_prepared_refresh()
            (res), ms_exp = timed(lambda: b.export_table())
        if r > 0:
            t_pure.append(ms_pure)
            t_hop.append([ms_walk, ms_exp])
            t_late.append(ms_late)
            n_merged.append(merged["batches_merged"])
            older = (table, n_table)
        else:
            exp0 = ms_exp
        n_table, buf = res
        table = buf[:max(n_table, 1) * L.TABLE_ENTRY_BYTES].clone()
        recs.append(n_table)
        del reads, batches
    ctx.close()
    exch = 2 * 2 * (tai / 8 / N) / (LINK * 1e9) * 1e3 * (N - 1) / max(N - 1, 1)     # two exchanges x (reduce-scatter + all-gather): tai/8/N bytes per link and phase
    p1 = max(p + l for p, l in zip(t_first, t_second)) + exch
    hops = 0.0
    for r in range(1, N):
        send = recs[r - 1] * L.TABLE_ENTRY_BYTES / (LINK * 1e9) * 1e3
        hops += send + t_hop[r - 1][0] + (t_hop[r - 1][1] if r < N - 1 else 0.0)
    p2 = max(t_scan0, max(t_pure) if t_pure else 0.0) + exp0 * (N > 1) + hops
    step = p1 + p2
    print(f"N={N}: per rank {per} reads | pass 1 ({'own load + fix-up' if fixup else 'presence + load'}): {min(t_first):.0f}-{max(t_first):.0f} ms + {min(t_second):.0f}-{max(t_second):.0f} ms + exchanges {exch:.0f} ms = {p1:.0f} ms | "
          f"pass 2: rank 0 scan {t_scan0:.0f} ms, others' pure stage {min(t_pure) if t_pure else 0:.0f}-{max(t_pure) if t_pure else 0:.0f} ms, hops (send + import + walk + export) "
          f"{' '.join(f'{recs[r - 1] * 32 / LINK / 1e6:.0f}+{t_hop[r - 1][0]:.0f}+{t_hop[r - 1][1]:.0f}' for r in range(1, N))} = {hops:.0f} ms | step {step:.0f} ms = {kmers / step * 1e3:.3g} k-mers/s "
          f"| fresher previews (off the chain): {' '.join(f'{x:.0f}' for x in t_late)} ms, batches whose planes the walk merged: {sum(n_merged)} "
          f"(records {recs[-1]}, junctions of the last rank's stats {stats['n_junctions']}; bloo1 and bloo2 of the shards' pass 1 {'EQUAL' if filters_ok else 'DIFFER FROM'} the oracle's digests)", flush=True)
print(f"(links priced at {LINK:.0f} GB/s per direction; the one-GPU step of the same workload is the bench line's full_size.config4)")
