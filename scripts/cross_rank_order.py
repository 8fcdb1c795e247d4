"""What would a per-KEY order of the junction walk across GPUs buy?  

The ordered walk (ReadScanner.cpp:61-231) is one sequence over all reads because every read reads and writes one junction map.  Its
per-key form -- a read waits only for the earlier reads that hold the same junction k-mer -- is what k_walk_ko does inside one window of
one GPU.  Before building its cross-rank form this script measures, on the bench's own data (config 2: 10 M reads of a 20 Mb genome, the
junction k-mers taken from the GPU scan itself), what that order leaves to overlap:

  1. contiguous file-order shards (north_star's layout): for every shard boundary, the lag rank r+1 must keep behind rank r so that no
     read of it ever waits -- as a fraction of rank r's shard.  1.0 = rank r+1 starts when rank r ends: the chain of section 5.
  2. a list-scheduling simulation of the walk (scripts/cross_rank_order_sim.c) over the real (read, junction k-mer) incidences: N ranks
     of W walkers, reads handed out in file order per rank, per-key order across all ranks, unit cost per read; contiguous shards against
     shards interleaved in units of g reads.

    gpurun -- python scripts/cross_rank_order.py [n_reads]
"""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from faucet_amd import api  # noqa: E402

K = 31
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
scale = n_reads / 10_000_000
dev = torch.device("cuda", 0)
reads = bench.make_reads(bench.make_genome(int(20_000_000 * scale), 2, dev), n_reads, 100, 0.01, 1000, dev)
tai, nh = api.load_filter_shape(int(100_000_000 * scale), int(20_000_000 * scale))
batches = bench.device_batches(reads, bench.batch_bounds(n_reads, 1_000_000, 2))
ctx = api.Context(K, tai, nh)
bench.step_single(ctx, batches)
keys, _ = ctx.junctions()
ctx.close()
print(f"{n_reads} reads, {len(keys)} junction records", flush=True)


def revcomp(km):
    out = torch.zeros_like(km)
    x = km
    for _ in range(K):
        out = (out << 2) | ((x & 3) ^ 2)
        x = x >> 2
    return out


jk = torch.from_numpy(keys.astype(np.int64)).to(dev)
canon_keys = torch.unique(torch.minimum(jk, revcomp(jk)))
n_keys = canon_keys.numel()
print(f"{n_keys} distinct canonical junction k-mers", flush=True)

code = torch.zeros(256, dtype=torch.int64, device=dev)
for i, c in enumerate(b"ACTG"):          # NT2int (utils/Kmer.cpp): A C T G = 0 1 2 3, complement = code ^ 2
    code[c] = i
npos = 100 - K + 1
occ_counts = np.zeros(n_reads, dtype=np.int64)
occ_keys = []
t0 = time.time()
for lo in range(0, n_reads, 500_000):
    c = code[reads[lo:lo + 500_000].long()]
    fw = torch.zeros((c.shape[0], npos), dtype=torch.int64, device=dev)
    rc = torch.zeros_like(fw)
    for i in range(K):
        fw = (fw << 2) | c[:, i:i + npos]
        rc = rc | ((c[:, i:i + npos] ^ 2) << (2 * i))
    canon = torch.minimum(fw, rc)
    idx = torch.searchsorted(canon_keys, canon).clamp_(max=n_keys - 1)
    hit = canon_keys[idx] == canon
    occ_counts[lo:lo + c.shape[0]] = hit.sum(1).cpu().numpy()
    occ_keys.append(idx[hit].to(torch.int32).cpu().numpy())       # row-major: in read order
occ_key = np.concatenate(occ_keys)
occ_start = np.zeros(n_reads + 1, dtype=np.int64)
np.cumsum(occ_counts, out=occ_start[1:])
occ_read = np.repeat(np.arange(n_reads, dtype=np.int64), occ_counts)
print(f"{len(occ_key)} (read, junction k-mer) incidences = {len(occ_key) / n_reads:.2f} per read, {len(occ_key) / n_keys:.1f} per k-mer "
      f"({time.time() - t0:.1f} s)", flush=True)

# ---- 1. contiguous shards: the lag a rank must keep behind its predecessor ---------------------------------------------------------------
print("\n== contiguous file-order shards: how far rank r must have come before rank r+1's reads stop waiting")
for n_ranks in (2, 8):
    bounds = np.linspace(0, n_reads, n_ranks + 1).astype(np.int64)
    rows = []
    for r in range(n_ranks - 1):
        a0, a1, b1 = bounds[r], bounds[r + 1], bounds[r + 2]
        last = np.full(n_keys, -1, dtype=np.int64)
        sel = slice(occ_start[a0], occ_start[a1])
        np.maximum.at(last, occ_key[sel], occ_read[sel])              # last read of shard r that holds the k-mer
        selb = slice(occ_start[a1], occ_start[b1])
        need = np.full(b1 - a1, -1, dtype=np.int64)                   # per read of shard r+1: the last read of shard r it waits for
        np.maximum.at(need, occ_read[selb] - a1, last[occ_key[selb]])
        frac_need = np.where(need >= 0, (need - a0 + 1) / (a1 - a0), 0.0)
        own = np.arange(b1 - a1) / (b1 - a1)
        lag = float(np.max(frac_need - own))                           # both ranks walking at the same pace
        first = frac_need[: max(1, (b1 - a1) // 100)]
        rows.append((r, lag, float(np.median(first)), float(np.quantile(first, 0.1)), float(first.max()), float((need >= 0).mean())))
    for r, lag, med, q10, mx, dep in (rows if len(rows) == 1 else rows[:1] + rows[-1:]):
        print(f"  {n_ranks} ranks, boundary {r}|{r + 1}: {100 * dep:.1f} % of rank {r + 1}'s reads hold a k-mer rank {r} also holds; its first 1 % of reads "
              f"wait for rank {r} to be {100 * q10:.1f} % (10th percentile) / {100 * med:.1f} % (median) / {100 * mx:.2f} % (max) through its shard; "
              f"lag without waiting = {lag:.4f} of a shard")

# ---- 2. list scheduling under the per-key order --------------------------------------------------------------------------------------------
so = "/tmp/cross_rank_order_sim.so"
subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(os.path.dirname(os.path.abspath(__file__)), "cross_rank_order_sim.c")], check=True)
lib = C.CDLL(so)
lib.simulate.restype = C.c_double
lib.simulate.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]


def simulate(rank_of, n_ranks, walkers):
    rank_of = np.ascontiguousarray(rank_of, dtype=np.int32)
    wait = C.c_double()
    ends = np.zeros(n_ranks, dtype=np.float64)
    span = lib.simulate(n_reads, occ_start.ctypes.data, occ_key.ctypes.data, n_keys, rank_of.ctypes.data, n_ranks, walkers, C.byref(wait), ends.ctypes.data)
    return span, wait.value


ar = np.arange(n_reads, dtype=np.int64)
print("\n== the walk as list scheduling under the per-key order (time unit = one read's walk; W walkers per rank; `waiting` = share of walker time spent holding a read that waits)")
for walkers in (1024, 8192):
    base, w0 = simulate(np.zeros(n_reads, dtype=np.int32), 1, walkers)
    print(f"  W = {walkers}: one rank {base:.0f} units (ideal {n_reads / walkers:.0f}; waiting {100 * w0 / (base * walkers):.1f} %)")
    for n_ranks in (2, 4, 8):
        span, w = simulate(ar * n_ranks // n_reads, n_ranks, walkers)
        line = f"    {n_ranks} ranks: contiguous shards {span:.0f} ({base / span:.2f}x, waiting {100 * w / (span * walkers * n_ranks):.0f} %)"
        for g in (100_000, 10_000, 1_000, 64):
            span, w = simulate((ar // g) % n_ranks, n_ranks, walkers)
            line += f" | interleaved by {g}: {span:.0f} ({base / span:.2f}x, {100 * w / (span * walkers * n_ranks):.0f} %)"
        print(line, flush=True)
