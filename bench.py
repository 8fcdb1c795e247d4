#!/usr/bin/env python3
"""bench.py — canonical k-mers/s (load+scan) of the MI355X path on BASELINE.json's config.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path (Bloom load pass + junction scan pass) over the whole synthetic read set,
which is resident in HBM before the timed region starts.  N = 1: BASELINE.json configs[1] — 10 M synthetic
100 bp reads, k = 31, -estimated_kmers 1e8 -singletons 2e7 (64 MiB filters, 3 hash functions).  N > 1 (default
`--scaling strong`): BASELINE.json configs[3], the configuration the 8-GPU target is stated on -- 200 M x 100 bp reads of a
400 Mb genome, -estimated_kmers 1e9 -singletons 2e8 (2 x 1 GiB filters) -- cut into N file-order shards (25 M reads per rank at
N = 8): the same reads as the N = 1 line's `full_size.config4` leg and as tests/golden/fullsize.json, so the line also says whether
bloo2 and the junction keys came out as the oracle's.  `--scaling weak` is the round 1-3 mode: every rank holds 10 M reads of an
N x 20 Mb genome and the filters are sized for N x 1e8 k-mers.  Either way reads are sharded in file order, the exclusive
prefix-OR of the shards' bloo1 and the OR-all-reduce of their bloo2 run as slice-wise reduce-scatter / all-gather (grouped RCCL
send/recv + local OR kernel), the pure scan stage runs on every rank at once and the ordered junction walk is handed from rank to
rank (table export -> send/recv -> import); `rank_stage_ms` lists every rank's stages of the last timed step.

The timed steps run WITHOUT HIP events around the kernels (they cost 1.5 % of a step, profiles/r04_profile_flag_ab.txt); kernel
times, `roofline` and `device_time_share` come from separate bracketed steps of the same context right behind them (`profiled_steps`).

The timed region of a step ends when the pass outputs are final in HOST memory: the bloo2 bit array and the
junction records in creation order.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from faucet_amd import _lib as L  # noqa: E402
from faucet_amd import api, sharded  # noqa: E402

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FALLBACKS = []      # steps in which the lazy-flag self-check fired and the scan was repeated with eager flags
METRIC = "canonical k-mers/s (load+scan) at k=31, 100bp reads; % HBM roofline"


# ---------------------------------------------------------------------------------------------- synthetic data
def make_genome(length, seed, device):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    return acgt[torch.randint(0, 4, (length,), generator=g, device=device)]


def make_reads(genome, n_reads, read_len, err, seed, device, chunk=1_000_000):
    """(n_reads, read_len) uint8 ASCII in HBM: uniform placement, i.i.d. substitutions, half reverse-complemented
    (same shape of data as faucet_amd/synth.py, generated on the device)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    G = genome.numel()
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    code = torch.zeros(256, dtype=torch.int64, device=device)
    code[acgt.long()] = torch.arange(4, device=device)
    comp = torch.zeros(256, dtype=torch.uint8, device=device)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    out = torch.empty((n_reads, read_len), dtype=torch.uint8, device=device)
    ar = torch.arange(read_len, device=device)
    for lo in range(0, n_reads, chunk):
        n = min(chunk, n_reads - lo)
        starts = torch.randint(0, G - read_len + 1, (n,), generator=g, device=device)
        r = genome[starts[:, None] + ar[None, :]]
        if err > 0:
            m = torch.rand((n, read_len), generator=g, device=device) < err
            shift = torch.randint(1, 4, (n, read_len), generator=g, device=device)
            r = torch.where(m, acgt[(code[r.long()] + shift) & 3], r)
        rc = torch.rand((n,), generator=g, device=device) < 0.5
        r = torch.where(rc[:, None], comp[r.long()].flip(1), r)
        out[lo:lo + n] = r
    return out


def batch_bounds(n, batch_reads, ramp):
    """[lo, hi) read ranges of the batches, in file order.  Results do not depend on the batching (tests/test_gpu_parity.py);
    its cost does: the first batches of both passes meet an empty carry / an empty junction map (everything pending, every
    junction test evaluated), and the walk of the last batch overlaps nothing.  With `ramp` the batches therefore grow from
    batch_reads / 2^ramp (the first two alike: the preview of batch b sees the map as of batch b-2) by doubling and shrink
    again at the end."""
    if ramp <= 0 or n < 4 * batch_reads:
        sizes = [batch_reads] * (n // batch_reads)
    else:
        first = max(batch_reads >> ramp, 1)
        head = [first] + [first << i for i in range(ramp)]            # f, f, 2f, ... batch/2
        tail = [batch_reads >> i for i in range(1, min(ramp, 3) + 1)]  # batch/2, batch/4, batch/8
        mid = n - sum(head) - sum(tail)
        body = [batch_reads] * (mid // batch_reads)
        rest = mid - sum(body)
        sizes = head + ([rest] if rest else []) + body + tail
    if sum(sizes) < n:
        sizes.append(n - sum(sizes))
    bounds, lo = [], 0
    for s in sizes:
        bounds.append((lo, lo + s))
        lo += s
    assert lo == n
    return bounds


def device_batches(reads, bounds):
    """ReadBatch views (device pointers) over consecutive row blocks of the read matrix"""
    n, ln = reads.shape
    if isinstance(bounds, int):                        # equal batches of that many reads
        bounds = batch_bounds(n, bounds, 0)
    offs = torch.arange(n + 1, dtype=torch.int64, device=reads.device) * ln
    out = []
    for lo, hi in bounds:
        o = offs[lo:hi + 1]
        out.append(api.ReadBatch(reads.data_ptr(), o.data_ptr(), n_reads=hi - lo, on_device=True, keepalive=(reads, offs, o),
                                 n_positions=(hi - lo) * (ln + 1)))
    # the library runs on its own non-blocking stream: what torch's stream is still writing (reads, offsets) must be finished
    torch.cuda.synchronize(reads.device)
    return out


# ---------------------------------------------------------------------------------------------- one step
def step_single(ctx, batches, pinned=False):
    """pinned=True (what main() times): the outputs land in page-locked buffers owned by the context -- bloo2 on a copy stream
    while the scan runs -- and are views that live as long as the context; pinned=False returns independent arrays."""
    ctx.load_begin()
    for b in batches:
        ctx.load_batch(b)
    lst = ctx.load_end()
    # pass-1 output on its way to host memory (page-locked) while the scan runs; the reference keeps bloo2 for the dump and for
    # Stage 3, the scan in between reads the device copy
    if pinned:
        bloo2 = ctx.bloom_download_begin(L.BLOO2, ctx._pinned_buffer("bloo2", ctx.tai // 8))
    else:
        bloo2 = ctx.bloom_download(L.BLOO2)
    # ReadScanner.scanReads = scan_begin / scan_batch... / scan_end, plus the documented reaction to a failed lazy-flag self-check
    # (DESIGN.md section 4): close the pass, switch to eager junction tests, scan again -- inside the timed region if it happens
    sc = api.ReadScanner(ctx)
    sst = sc.scanReads(batches)
    if sc.fell_back_to_eager:
        FALLBACKS.append(1)
    keys, recs = ctx.junctions(pinned=pinned)       # pass-2 output final in host memory (creation order)
    ctx.bloom_download_wait()                       # ... and so is pass 1's
    return lst, sst, bloo2, keys, recs


def step_multi(shard, batches, rank, world):
    """The read-sharded pipeline of faucet_amd/sharded.py on this rank's GPU."""
    lst = sharded.load_sharded(shard, batches, rank, world)
    ctx = shard.ctx
    # pass-1 output on its way to (page-locked) host memory while rank 0 starts the scan: it heads the chain of walks
    bloo2 = ctx.bloom_download_begin(L.BLOO2, ctx._pinned_buffer("bloo2", ctx.tai // 8)) if rank == 0 else None
    sst, last = sharded.scan_sharded(shard, batches, rank, world)
    keys = recs = None
    if last:
        keys, recs = ctx.junctions(pinned=True)                            # pass-2 output final in host memory
    ctx.bloom_download_wait()
    return lst, sst, bloo2, keys, recs


# ---------------------------------------------------------------------------------------------- CPU legs (oracle)
def cpu_baseline(reads_host, k, tai, nh):
    """The oracle ("port" of the reference's single-threaded path) timed on this host: load + scan of a bounded
    sample with the SAME filter size as the GPU run.  Returns (k-mers/s, seconds, n_kmers)."""
    from oracle import pyoracle as po
    bases, offs = po.reads_from_matrix(reads_host)
    b1, b2 = po.Bloom(tai, nh), po.Bloom(tai, nh)
    t0 = time.perf_counter()
    lst = po.load_two_filters(b1, b2, bases, offs, k)
    sc = po.Scanner(k, 1, 100, b2)
    sc.scan_reads(bases, offs)
    dt = time.perf_counter() - t0
    return lst.kmers / dt, dt, int(lst.kmers)


def _cpu_replica(args):
    """one worker of cpu_all_cores: the oracle on its own shard of the sample, with filters of the run's size"""
    reads_host, k, tai, nh = args
    v, dt, nk = cpu_baseline(reads_host, k, tai, nh)
    return nk, dt


def cpu_all_cores(reads_host, k, tai, nh, workers):
    """SURVEY 8d(ii)'s second CPU figure.  The reference is single-threaded and so is the oracle; what an ideal parallel port could do on
    this host is bounded by `workers` independent single-threaded replicas, each running load + scan over its own slice of the sample with
    its own full-size filters (no exchange between them: NOT the reference's result for the whole sample, only its work rate).  Returns
    (aggregate k-mers/s, wall seconds, k-mers)."""
    import multiprocessing as mp
    n = reads_host.shape[0]
    per = n // workers
    if per < 1000:
        return None
    parts = [(np.ascontiguousarray(reads_host[i * per:(i + 1) * per]), k, tai, nh) for i in range(workers)]
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(workers) as pool:      # spawn: the children must not inherit this process' GPU state
        res = pool.map(_cpu_replica, parts)
    dt = time.perf_counter() - t0
    nk = sum(r[0] for r in res)
    return nk / max(max(r[1] for r in res), 1e-9), dt, nk


def reference_binary_baseline(reads_host, k, E, S):
    """The COMPILED REFERENCE itself (oracle/_ref/faucet_ref, built from the reference's own sources by oracle/Makefile where
    they are mounted; it travels to the GPU box as a prebuilt binary) on a bounded sample written to a FASTA file, with the
    full-size filters of the run.  Its passes are timed from the moments its own progress lines appear on an unbuffered
    stdout ("Weights before load" .. "Weights after load", "Weight before read scan" .. "Reads processed" of the scan); the
    process is stopped once the junction file is written (its contig-graph stage is not part of the metric).
    Returns (k-mers/s, load s, scan s) or None when the binary is not there or does not behave."""
    import shutil
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "oracle", "_ref", "faucet_ref")
    if not os.path.exists(exe) or shutil.which("stdbuf") is None:
        return None
    # under a profiler the child would inherit its preload: stdbuf would initialise the GPU and then exec the reference, the
    # exec hop this pool forbids.  The port's number stands in then.
    if any(k.startswith(("ROCP", "ROCPROF", "ROCTRACER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None
    n, ln = reads_host.shape
    td = tempfile.mkdtemp(prefix="faucet_ref_")
    try:
        rec = np.empty((n, 10 + ln + 1), dtype=np.uint8)
        rec[:, 0] = ord(">")
        idx = np.arange(n, dtype=np.int64)
        for d in range(8):
            rec[:, 8 - d] = ord("0") + (idx // 10 ** d) % 10
        rec[:, 9] = ord("\n")
        rec[:, 10:10 + ln] = reads_host
        rec[:, 10 + ln] = ord("\n")
        fa = os.path.join(td, "sample.fa")
        rec.tofile(fa)
        cmd = ["stdbuf", "-o0", exe, "-read_load_file", fa, "-read_scan_file", fa, "-size_kmer", str(k), "-max_read_length", str(ln),
               "-estimated_kmers", str(E), "-singletons", str(S), "-file_prefix", os.path.join(td, "out"), "--no_cleaning"]
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        marks, buf = {}, b""
        t_start = time.perf_counter()
        while True:
            chunk = p.stdout.read1(65536) if hasattr(p.stdout, "read1") else p.stdout.read(1)
            now = time.perf_counter()
            if not chunk:
                break
            buf += chunk
            for key, pat in (("load0", b"Weights before load"), ("load1", b"Weights after load"), ("scan0", b"Weight before read scan"),
                             ("scan1", b"Time in seconds for read scan"), ("done", b"Done writing to junction file")):
                if key not in marks and pat in buf:
                    marks[key] = now
            if "done" in marks or now - t_start > 600:
                break
        p.kill()
        p.wait()
        if not all(m in marks for m in ("load0", "load1", "scan0", "scan1")):
            return None
        t_load, t_scan = marks["load1"] - marks["load0"], marks["scan1"] - marks["scan0"]
        return n * (ln - k + 1) / (t_load + t_scan), t_load, t_scan
    except Exception:
        return None
    finally:
        shutil.rmtree(td, ignore_errors=True)


def reference_bit_counts(k, read_len, err, coverage, bits_per_kmer_ratio, seed=77):
    """Bit accesses per k-mer that the REFERENCE semantics perform (early exit and skipping included), counted by
    the oracle on a scaled-down read set with the bench's coverage, error rate and filter bits per estimated k-mer.
    Returns dict(T_load, T_valid, T_junc, rho)."""
    from faucet_amd import synth
    from oracle import pyoracle as po
    tai = 1 << 24
    E = int(tai / bits_per_kmer_ratio)
    n_reads = int(E * 0.1)                  # bench: 1e7 reads for E = 1e8
    G = int(n_reads * read_len / coverage)
    g = synth.make_genome(G, seed)
    r = synth.make_reads(g, n_reads, read_len, err, seed + 1)
    bases, offs = po.reads_from_matrix(r)
    b1, b2 = po.Bloom(tai, 3), po.Bloom(tai, 3)
    lst = po.load_two_filters(b1, b2, bases, offs, k)
    b2.reset_counters()
    sc = po.Scanner(k, 1, 100, b2)
    sc.scan_reads(bases, offs)
    tests, _ = b2.counters()
    tv = sc.bit_tests_valid()
    n = lst.kmers
    rho = lst.to_bloo2 / n
    return {"T_load": 3 + 3 * rho, "T_valid": tv / n, "T_junc": (tests - tv) / n, "rho": rho, "sample_reads": n_reads, "sample_genome": G}


# ---------------------------------------------------------------------------------------------- main
def cli_leg(reads, args, kmers, want_junctions, variants=()):
    """writes the reads as FASTA (fixed-width headers), runs faucet_amd/faucet on it twice, reports the faster run; `variants`: (key, extra
    arguments) pairs run on the same file afterwards (the C++ host over read shards: `-gpus N`), reported under their key"""
    import re
    import shutil
    import tempfile
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "faucet_amd", "faucet")
    host = reads.cpu().numpy()
    n, L_ = host.shape
    d = tempfile.mkdtemp(prefix="faucet_bench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        rec = np.empty((n, 10 + L_ + 1), dtype=np.uint8)
        rec[:, 0] = ord(">")
        idx = np.arange(n, dtype=np.int64)
        for dgt in range(8):
            rec[:, 8 - dgt] = ord("0") + (idx // 10 ** dgt) % 10
        rec[:, 9] = ord("\n")
        rec[:, 10:10 + L_] = host
        rec[:, 10 + L_] = ord("\n")
        path = os.path.join(d, "reads.fa")
        rec.tofile(path)
        size = os.path.getsize(path)
        del rec
        cmd = [exe, "-read_load_file", path, "-read_scan_file", path, "-size_kmer", str(args.k), "-max_read_length", str(L_),
               "-estimated_kmers", str(args.estimated_kmers), "-singletons", str(args.singletons), "--no_cleaning",
               "-file_prefix", os.path.join(d, "out")]
        best = None
        parent_cpu = []          # CPU seconds THIS process burnt while the child ran (its threads should be asleep: the job's CPU quota is shared)
        for _ in range(2):
            t0 = time.perf_counter()
            c0 = time.process_time()
            r = subprocess.run(cmd, capture_output=True, text=True)
            dt = time.perf_counter() - t0
            parent_cpu.append(round(time.process_time() - c0, 3))
            if r.returncode != 0:
                raise RuntimeError("faucet exited with %d: %s" % (r.returncode, r.stderr[-300:]))
            best = dt if best is None else min(best, dt)
        m = re.search(r"Distinct junctions: (\d+)", r.stdout)
        import hashlib
        digest = lambda ext: hashlib.sha256(open(os.path.join(d, "out." + ext), "rb").read()).hexdigest()   # noqa: E731
        want_files = {ext: digest(ext) for ext in ("bloom", "junctions")}
        extra = {}
        for key, more in variants:
            vb = None
            for _ in range(2):
                t0 = time.perf_counter()
                rv = subprocess.run(cmd[:-2] + ["-file_prefix", os.path.join(d, key)] + list(more), capture_output=True, text=True)
                dtv = time.perf_counter() - t0
                if rv.returncode != 0:
                    extra[key] = {"error": "faucet exited with %d: %s" % (rv.returncode, rv.stderr[-300:])}
                    break
                vb = dtv if vb is None else min(vb, dtv)
            else:
                same = {ext: hashlib.sha256(open(os.path.join(d, key + "." + ext), "rb").read()).hexdigest() == want_files[ext] for ext in want_files}
                extra[key] = {"arguments": list(more), "seconds": vb, "value": kmers / vb, "unit": "k-mers/s", "files_equal_the_single_device_runs": same,
                              "note": "the same file through the C++ host that shards the reads (faucet_amd/host/shard_host.h): one host thread and one context per "
                                      "shard, ALL on this box's one device -- a functional record of the N-GPU command line, not a scaling number"}
        return {"seconds": best, "value": kmers / best, "unit": "k-mers/s", "input_bytes": size, **extra,
                "output_bytes": os.path.getsize(os.path.join(d, "out.bloom")) + os.path.getsize(os.path.join(d, "out.junctions")),
                "junctions_equal_the_steps": bool(m) and int(m.group(1)) == int(want_junctions),
                "parent_cpu_seconds_during_the_runs": parent_cpu, "load_average": list(os.getloadavg()),
                "note": "wall time of the whole `faucet` process (runtime start-up, both passes reading the FASTA file from page cache or tmpfs, "
                        ".bloom and .junctions written), best of two runs"}
    finally:
        shutil.rmtree(d, ignore_errors=True)



def _fullsize_traffic(config, kernel):
    """HBM bytes per launch (FETCH_SIZE + WRITE_SIZE) of `kernel` on a full-size configuration, from the committed counter passes; None if absent"""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            ks = json.load(f)["full_size"][config]["kernels"]
        for n, v in ks.items():
            if n.split("<")[0] == kernel:
                return float(v["bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        pass
    return None


def full_size_leg(name, device, batch_reads):
    """One of BASELINE.json's other configurations at its FULL size as one cold step (load + scan, reads resident in HBM, junctions and
    bloo2 back in host memory), on the reads of the committed parity fixture (tests/golden/fullsize.json: faucet_amd/synth_det.py's
    counter-based generator), so that the step's counters can be compared with the oracle's on the spot.  The digests themselves are
    checked by tests/test_gpu_fullsize.py; this leg puts a driver-timed rate for the configuration into the bench line."""
    from faucet_amd import synth_det as sd
    with open(os.path.join(ROOT, "tests", "golden", "fullsize.json")) as f:
        fx = json.load(f).get(name)
    if fx is None:
        return {"skipped": f"no fixture named {name} in tests/golden/fullsize.json"}
    c = fx["params"]
    t0 = time.perf_counter()
    reads = sd.make_reads(sd.make_genome(c["genome"], c["genome_seed"], device), c["reads"], c["read_len"], c["err"], c["read_seed"], device)
    tai, nh = api.load_filter_shape(c["E"], c["S"])
    ctx = api.Context(c["k"], tai, nh, device=device.index or 0, profile=True)
    batches = device_batches(reads, batch_bounds(c["reads"], batch_reads, 2))
    torch.cuda.synchronize(device)
    setup_seconds = time.perf_counter() - t0
    kmers = c["reads"] * (c["read_len"] - c["k"] + 1)
    steps = []
    for _ in range(2):          # the first use of a context (window calibration, junction table grown from its initial size), then a second step
        ctx.kernel_times_reset()
        t1 = time.perf_counter()
        lst, sst, bloo2, keys, recs = step_single(ctx, batches, pinned=True)
        ctx.synchronize()
        dt = time.perf_counter() - t1
        same = {k2: int(sst[k2]) == int(v) for k2, v in fx["counters"].items() if k2 in sst}
        same["to_bloo2"] = int(lst["to_bloo2"]) == int(fx["to_bloo2"])
        same["junction_records"] = len(keys) == int(fx["counters"]["n_junctions"])
        all_kt = ctx.kernel_times()
        kt = sorted(all_kt.items(), key=lambda kv: -kv[1][1])[:6]
        lm = all_kt.get("load_mark")
        roof = None
        if lm and lm[0]:
            avg_ms = lm[1] / lm[0]
            achieved = 64.0 * nh * (kmers / lm[0]) / (avg_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": "load_mark", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                    "avg_launch_ms": avg_ms, "launches": lm[0], "algorithmic_bytes_per_kmer": 64.0 * nh, "kmers_per_launch": kmers / lm[0]}
            # the counters' view of the same kernel on this configuration (round 6: scripts/profile_fullsize.sh <config> r06 pmc -> profiles/pmc_traffic.json)
            traffic = _fullsize_traffic(name, "k_load_mark")
            roof["traffic"] = traffic
            roof["frac_measured_traffic"] = traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS if traffic else None
            roof["traffic_over_algorithmic"] = traffic / (64.0 * nh * kmers / lm[0]) if traffic else None
        steps.append({"seconds": dt, "value": kmers / dt, "counters_equal_the_oracles": all(same.values()),
                      "differing": sorted(k2 for k2, v in same.items() if not v), "kernel_ms": {n: round(ms, 1) for n, (cnt, ms) in kt}, "roofline_load_mark": roof,
                      "junction_tests_run_by_the_walk": int(sst["flags_filled"]), "late_junction_tests": ctx.diag_late_flags(),
                      "scan_replays_so_far": ctx.diag_scan_replays()})
    out = {"seconds": steps[1]["seconds"], "value": steps[1]["value"], "unit": "k-mers/s", "kmers": kmers, "junctions": int(len(keys)),
           "counters_equal_the_oracles": all(st["counters_equal_the_oracles"] for st in steps),
           "first_step_of_the_context": steps[0], "second_step": steps[1], "setup_seconds": setup_seconds,
           "workload": f"{c['reads']} x {c['read_len']} bp of a {c['genome']} bp genome, {100 * c['err']:.0f} % substitutions, k={c['k']}, estimated_kmers={c['E']}, "
                       f"singletons={c['S']}: filters 2 x {tai // 8 >> 20} MiB, {nh} hash functions",
           "note": "two whole steps (load + scan, reads resident in HBM, bloo2 and the junctions back in host memory): the first use of the context "
                   "(window calibration, junction table grown from its initial size) and a second one, which `value` quotes; counters, to_bloo2 "
                   "and the junction count of both compared with the oracle's in tests/golden/fullsize.json"}
    ctx.close()
    del reads, batches
    torch.cuda.empty_cache()
    return out


def config3_cli_leg(device):
    """BASELINE config 3's shape at its real size -- 2.5 M pairs of 100-base reads of a 4.6 Mb genome with planted repeats, interleaved FASTQ,
    `--fastq --paired_ends` with cleaning -- file to files through faucet_amd/faucet: the configuration on which the path is SLOWEST (the
    ordered walk of the repeat clusters).  The reads are the deterministic ones of tests/golden/fullsize.json (config3), so
    the four files the CLI writes are compared, by digest, with what the COMPILED REFERENCE wrote on the same text in the build container."""
    import hashlib
    import re
    import shutil
    import tempfile
    from faucet_amd import synth_det as sd
    with open(os.path.join(ROOT, "tests", "golden", "fullsize.json")) as f:
        fx = json.load(f)["config3"]
    c = fx["params"]
    g = sd.make_genome(c["genome"], c["genome_seed"], device)
    sd.plant_repeats(g, c["genome_seed"] + 100, *c["repeats"])
    reads = sd.make_pairs(g, c["pairs"], c["read_len"], c["insert"][0], c["insert"][1], c["err"], c["read_seed"], device)
    text = sd.fasta_bytes(reads, fastq=True).cpu().numpy()
    kmers = int(reads.shape[0]) * (c["read_len"] - c["k"] + 1)
    del reads, g
    exe = os.path.join(ROOT, "faucet_amd", "faucet")
    d = tempfile.mkdtemp(prefix="faucet_bench3_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        path = os.path.join(d, "reads.fq")
        text.tofile(path)
        size = os.path.getsize(path)
        del text
        cmd = [exe, "-read_load_file", path, "-read_scan_file", path, "-file_prefix", os.path.join(d, "out")] + fx["args"]
        best, phases, notes = None, None, []
        parent_cpu = []
        for _ in range(2):
            t0 = time.perf_counter()
            c0 = time.process_time()
            r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, FGPU_CLI_TIMES="1"))
            dt = time.perf_counter() - t0
            parent_cpu.append(round(time.process_time() - c0, 3))
            if r.returncode != 3:          # cleaning is on: "outputs written, contig graph not built"
                raise RuntimeError("faucet exited with %d: %s" % (r.returncode, r.stderr[-300:]))
            if best is None or dt < best:
                best = dt
                phases = {m.group(1).strip(): float(m.group(2)) for m in re.finditer(r"\[cli\] (pass [12][^\d]*?)\s+([0-9.]+) ms", r.stderr)}
                notes = [ln.strip()[6:].strip() for ln in r.stderr.splitlines() if "walked optimistically" in ln or "long pair filter on the device" in ln]
        m = re.search(r"Distinct junctions: (\d+)", r.stdout)

        def sha(pth):
            h = hashlib.sha256()
            with open(pth, "rb") as f:
                for blk in iter(lambda: f.read(1 << 24), b""):
                    h.update(blk)
            return h.hexdigest()

        same = {ext: sha(os.path.join(d, "out." + ext)) == fx[ext + "_sha256"] for ext in ("bloom", "junctions", "short_pair_filter", "long_pair_filter")}
        p12 = [v for n, v in (phases or {}).items() if n.startswith("pass 1")] + [v for n, v in (phases or {}).items() if n.startswith("pass 2")]
        return {"seconds": best, "value": kmers / best, "unit": "k-mers/s", "kmers": kmers, "input_bytes": size, "pass_ms": phases, "walks": notes, "parent_cpu_seconds_during_the_runs": parent_cpu, "load_average": list(os.getloadavg()),
                # SURVEY 8d's definition of the metric: N / (t_load + t_scan), each pass from its first input byte to its outputs final in host
                # memory -- the CLI's own clock around the two passes (file reading included), without HIP start-up, file dumps and process exit
                "load_scan_value": kmers / (sum(p12) / 1e3) if len(p12) == 2 else None,
                "junctions": int(m.group(1)) if m else None, "junctions_equal_the_references": bool(m) and int(m.group(1)) == fx["distinct_junctions"],
                "files_equal_the_references": same,
                "note": "wall time of the whole `faucet --fastq --paired_ends` process (start-up, both passes over a 1.07 GB interleaved FASTQ file in tmpfs, "
                        ".bloom, .junctions and both pair filters written), best of two runs; expected digests: the compiled reference on the same text"}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher around it: this process becomes the PARENT of the N ranks.  It has made no GPU call
    (importing torch and counting devices does not start the HIP runtime), starts `python -m torch.distributed.run` as a CHILD (never exec),
    relays the ranks' output -- rank 0's ONE JSON line on stdout -- and returns the child's exit code."""
    share = os.environ.get("FAUCET_SHARE_GPU", "0") == "1"
    seen = torch.cuda.device_count()
    if not share and seen < n:
        sys.stderr.write(f"bench.py --gpus {n}: only {seen} GPU(s) visible on this node (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES?); "
                         f"nothing was started.  (FAUCET_SHARE_GPU=1 FAUCET_DIST_BACKEND=gloo runs the N ranks on one device: functional check only)\n")
        return 2
    import socket
    with socket.socket() as s:              # a free port of this host: two benches side by side must not meet
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "4"))
    def attempt(argv, env, limit_s):
        """one `torch.distributed.run` child in a session of its own; past `limit_s` its whole process group (exactly the processes started
        here) is ended.  Returns the exit code, or None for "ran out of time"."""
        import signal
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + argv
        sys.stderr.write("[bench] starting %d ranks: %s\n" % (n, " ".join(cmd)))
        p = subprocess.Popen(cmd, env=env, start_new_session=True)
        try:
            return p.wait(timeout=limit_s if limit_s > 0 else None)
        except subprocess.TimeoutExpired:
            for sig in (signal.SIGTERM, signal.SIGKILL):
                try:
                    os.killpg(p.pid, sig)
                except ProcessLookupError:
                    break
                try:
                    p.wait(timeout=20)
                    break
                except subprocess.TimeoutExpired:
                    continue
            return None

    # RCCL with more than one rank has never run where this was built (one GPU per box).  So that a first contact with an N-GPU node
    # is not lost to a transport problem, a run over RCCL that fails or hangs (FAUCET_BENCH_RANKS_TIMEOUT seconds, default 450: a run of 8 ranks takes about 2 minutes with a cold start; 0 = no limit) is
    # followed by ONE run of the same protocol over gloo (device buffers staged through page-locked host memory), with few steps and the reason in
    # the line (`transport_fallback`): a functional record of the N-rank pipeline, not the xGMI number.  FAUCET_BENCH_NO_FALLBACK=1: off.
    limit = float(os.environ.get("FAUCET_BENCH_RANKS_TIMEOUT", "450"))
    rc = attempt(sys.argv[1:], env, limit)
    if rc == 0 or env.get("FAUCET_DIST_BACKEND", "nccl") != "nccl" or os.environ.get("FAUCET_BENCH_NO_FALLBACK") == "1":
        return 1 if rc is None else rc
    why = "no result after %.0f s" % limit if rc is None else "exit code %d" % rc
    sys.stderr.write(f"[bench] the run over RCCL ended with {why}; running the same pipeline over gloo (host-staged) as a functional record\n")
    argv, skip = [], False
    for a in sys.argv[1:]:           # the second run is short: 2 timed steps behind 1 warm-up, no CPU legs
        if skip:
            skip = False
        elif a in ("--steps", "--warmup"):
            skip = True
        elif not a.startswith(("--steps=", "--warmup=")):
            argv.append(a)
    argv += ["--steps", "2", "--warmup", "1", "--no-cpu", "--no-ceilings"]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    rc2 = attempt(argv, dict(env, FAUCET_DIST_BACKEND="gloo", FAUCET_BENCH_FALLBACK_REASON=f"backend nccl (RCCL) with {n} ranks: {why}"), 0)
    return 1 if rc2 is None else rc2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--fixture", default="config4", help="--scaling strong: whose reads (seeds) and default totals, an entry of tests/golden/fullsize.json")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="N > 1: strong (default) = BASELINE config 4's 200 M reads cut into N file-order shards; weak = 10 M reads per rank. "
                         "With strong, --reads / --genome / --estimated-kmers / --singletons are TOTALS (defaults: config 4's)")
    ap.add_argument("--reads", type=int, default=None, help="reads per GPU (weak; default 10 M) / in all (strong; default 200 M)")
    ap.add_argument("--read-len", type=int, default=100)
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--genome", type=int, default=None, help="genome bases per GPU (weak; default 20 Mb) / in all (strong; default 400 Mb)")
    ap.add_argument("--estimated-kmers", type=int, default=None, help="per GPU (weak; default 1e8) / in all (strong; default 1e9)")
    ap.add_argument("--singletons", type=int, default=None, help="per GPU (weak; default 2e7) / in all (strong; default 2e8)")
    ap.add_argument("--err", type=float, default=0.01)
    ap.add_argument("--batch-reads", type=int, default=None, help="reads per device call (default 1 M; 2.5 M with --scaling strong)")
    ap.add_argument("--ramp", type=int, default=int(os.environ.get("FAUCET_BENCH_RAMP", "2")),
                    help="grow the first batches from batch_reads / 2^RAMP by doubling and shrink the last ones (0 = equal batches)")
    ap.add_argument("--cpu-sample-reads", type=int, default=1_000_000)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU legs (cpu_baseline and reference bit counts)")
    ap.add_argument("--no-ceilings", action="store_true", help="skip the streaming-copy / random-access ceiling measurements")
    ap.add_argument("--profile-walk", action="store_true", help="time the per-window walk kernels individually")
    ap.add_argument("--host-input", action="store_true",
                    help="hand the reads over as HOST buffers (PCIe copy inside the timed region); diagnostic only, never the headline value")
    ap.add_argument("--no-full-size", action="store_true", help="skip the full-size legs of BASELINE configs 5 and 4 (N = 1; about half a minute)")
    ap.add_argument("--no-host-leg", action="store_true", help="skip the extra PCIe-inclusive steps reported as `host_input` (N = 1)")
    ap.add_argument("--cpu-workers", type=int, default=0, help="replicas of the all-cores CPU leg (0 = min(host cores, 16))")
    ap.add_argument("--no-profile", action="store_true",
                    help="timed steps WITHOUT HIP events around every kernel (A/B of what FGPU_FLAG_PROFILE costs; no roofline / kernel times in the line)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))

    # stdout carries exactly ONE JSON line: libraries that chat on fd 1 (RCCL prints its version banner there) go to stderr
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(line):
        os.write(json_fd, (line + "\n").encode())

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch sees no GPU (there is no CPU fallback)")
    if args.profile_walk:
        os.environ["FGPU_PROFILE_WALK"] = "1"
    # FAUCET_SHARE_GPU=1 + FAUCET_DIST_BACKEND=gloo: all ranks on cuda:0 with gloo as the transport -- a way to run the N > 1 code
    # path on a single-GPU box (functional check only; never a scaling number)
    if os.environ.get("FAUCET_SHARE_GPU", "0") == "1":
        local_rank = 0
    backend = os.environ.get("FAUCET_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # FAUCET_FORCE_SHARDED=1 runs the multi-GPU code path (RCCL collectives, table hand-over) even with one rank, so that
    # it can be exercised on a single-GPU box; it is never the default
    force_sharded = os.environ.get("FAUCET_FORCE_SHARDED", "0") == "1"
    if world > 1 or force_sharded:
        if "RANK" not in os.environ:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group(backend, rank=0, world_size=1, **({"device_id": device} if backend == "nccl" else {}))
        else:
            dist.init_process_group(backend, **({"device_id": device} if backend == "nccl" else {}))

    k, L_ = args.k, args.read_len
    strong = (args.scaling or ("strong" if world > 1 else "weak")) == "strong"
    fixture4 = None
    if strong:
        # BASELINE config 4, the reads of tests/golden/fullsize.json (faucet_amd/synth_det.py: a row is a function of (seed, row), so every rank
        # makes its own file-order shard of the ONE read set); other totals may be given for functional runs at reduced size
        from faucet_amd import synth_det as sd
        with open(os.path.join(ROOT, "tests", "golden", "fullsize.json")) as f:
            fixture4 = json.load(f)[args.fixture]
        c4 = fixture4["params"]
        total_reads = args.reads or c4["reads"]
        total_genome = args.genome or c4["genome"]
        E, S = args.estimated_kmers or c4["E"], args.singletons or c4["S"]
        if (total_reads, total_genome, E, S, L_, k, args.err) != (c4["reads"], c4["genome"], c4["E"], c4["S"], c4["read_len"], c4["k"], c4["err"]):
            fixture4 = None                                   # not the fixture's workload: nothing to compare digests with
        shard_lo, shard_hi = total_reads * rank // world, total_reads * (rank + 1) // world
        args.reads = shard_hi - shard_lo                      # (this rank's; `total_reads` is the job's)
        args.batch_reads = args.batch_reads or 2_500_000
        genome = sd.make_genome(total_genome, c4["genome_seed"], device)
        reads = sd.make_reads(genome, args.reads, L_, args.err, c4["read_seed"], device, first_row=shard_lo)
        args.genome, args.estimated_kmers, args.singletons = total_genome, E, S
    else:
        args.reads = args.reads or 10_000_000
        args.genome = args.genome or 20_000_000
        args.estimated_kmers = args.estimated_kmers or 100_000_000
        args.singletons = args.singletons or 20_000_000
        args.batch_reads = args.batch_reads or 1_000_000
        total_reads, total_genome = args.reads * world, args.genome * world
        E, S = args.estimated_kmers * world, args.singletons * world
        genome = make_genome(total_genome, 2, device)
        reads = make_reads(genome, args.reads, L_, args.err, 1000 + rank, device)
    tai, nh = api.load_filter_shape(E, S)
    del genome
    torch.cuda.empty_cache()
    bounds = batch_bounds(args.reads, args.batch_reads, args.ramp)
    if args.host_input:
        host = reads.cpu().numpy()
        batches = [api.ReadBatch.from_matrix(host[lo:hi]) for lo, hi in bounds]
    else:
        batches = device_batches(reads, bounds)
    torch.cuda.synchronize()

    if world > 1 or force_sharded:
        # the library on the stream the collectives are ordered with: kernels, local OR steps and RCCL calls follow each other on the device,
        # the host never waits between them (sharded.GpuShard.fence is a no-op then)
        tstream = torch.cuda.Stream(device)
        torch.cuda.set_stream(tstream)
        ctx = api.Context(k, tai, nh, device=local_rank, profile=False, walk_window_span=int(os.environ.get("FAUCET_WALK_SPAN", "0")), stream=tstream.cuda_stream)
        shard = sharded.GpuShard(ctx, device)
    else:
        ctx = api.Context(k, tai, nh, device=local_rank, profile=False, walk_window_span=int(os.environ.get("FAUCET_WALK_SPAN", "0")))
        shard = None

    def one_step():
        if world == 1 and not force_sharded:
            return step_single(ctx, batches, pinned=True)
        sharded.CLOCK.enable(True)                 # stage boundaries of this step as events on the stream (no synchronisation)
        return step_multi(shard, batches, rank, world)

    def fence():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    ctx.kernel_times_reset()
    fence()
    t0 = time.perf_counter()
    out = None
    for _ in range(args.steps):
        out = one_step()
    fence()
    elapsed = time.perf_counter() - t0
    lst, sst, bloo2, keys, recs = out
    kmers_local = lst["kmers"]
    stage_ms = sharded.CLOCK.report() if (world > 1 or force_sharded) else None      # this rank's stages of the last timed step
    sharded.CLOCK.enable(False)
    # ---- did the sharded run produce the sequential run's outputs?  (strong scaling on the fixture's reads: digests of tests/golden/fullsize.json)
    checks = None
    if strong:
        import hashlib
        mine = {}
        if bloo2 is not None:          # (rank 0 holds pass 1's output, the last rank pass 2's)
            mine["bloo2_sha256"] = hashlib.sha256(np.ascontiguousarray(bloo2)).hexdigest()
            if fixture4 is not None:
                mine["bloo2_equals_the_oracles"] = mine["bloo2_sha256"] == fixture4["bloo2_sha256"]
        if keys is not None:
            mine["junction_keys_sha256"] = hashlib.sha256(np.ascontiguousarray(keys)).hexdigest()
            mine["junction_records_sha256"] = hashlib.sha256(np.ascontiguousarray(recs)).hexdigest()
            mine["junctions"] = int(len(keys))
            if fixture4 is not None:
                mine["junction_keys_equal_the_oracles"] = (len(keys) == int(fixture4["counters"]["n_junctions"]) and
                                                           mine["junction_keys_sha256"] == fixture4["keys_sha256"])
                mine["junction_records_equal_the_oracles"] = mine["junction_records_sha256"] == fixture4["recs_sha256"]
                mine["scan_counters_equal_the_oracles"] = all(int(sst[c]) == int(v) for c, v in fixture4["counters"].items() if c in sst)
        checks = mine
    # ---- kernel times from separate, bracketed steps (the timed ones above ran without events)
    prof_steps, prof_elapsed = 0, 0.0
    if not args.no_profile:
        prof_steps = max(1, min(args.steps, 5))
        ctx.profile_enable(True)
        ctx.kernel_times_reset()
        fence()
        t1 = time.perf_counter()
        for _ in range(prof_steps):
            out = one_step()
        fence()
        prof_elapsed = time.perf_counter() - t1
        sharded.CLOCK.enable(False)
        lst, sst, bloo2, keys, recs = out
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        kk = torch.tensor([kmers_local], dtype=torch.int64, device=device)
        dist.all_reduce(kk, op=dist.ReduceOp.SUM)
        kmers_total = int(kk.item())
    else:
        kmers_total = kmers_local
    ktimes = ctx.kernel_times()
    if prof_steps:
        ctx.profile_enable(False)
    all_stage_ms = all_checks = None
    free_b, total_b = torch.cuda.mem_get_info(device)
    hbm = {"rank": rank, "device": local_rank, "used_bytes_after_the_steps": int(total_b - free_b), "total_bytes": int(total_b),
           "torch_peak_allocated_bytes": int(torch.cuda.max_memory_allocated(device))}
    all_hbm = [hbm]
    if world > 1:
        all_stage_ms, all_checks, all_hbm = [None] * world, [None] * world, [None] * world
        dist.all_gather_object(all_stage_ms, stage_ms)
        dist.all_gather_object(all_checks, checks)
        dist.all_gather_object(all_hbm, hbm)
    else:
        all_stage_ms, all_checks = ([stage_ms] if force_sharded else None), [checks]

    if rank != 0:
        dist.barrier()
        dist.destroy_process_group()
        return

    value = kmers_total * args.steps / elapsed
    res = {
        "metric": METRIC, "value": value, "unit": "k-mers/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic" + (" (host buffers: PCIe copies inside the timed region)" if args.host_input else ""),
        "config": {"workload": (f"BASELINE {args.fixture.replace('config', 'config ')}: " if strong and fixture4 else "") +
                               f"{total_reads} synthetic {L_} bp reads ({args.reads} per GPU), k={k}, "
                               f"estimated_kmers={E}, singletons={S}, genome {total_genome} bp, {args.err:.0%} substitutions; "
                               f"filters 2 x {tai // 8 >> 20} MiB, {nh} hash functions",
                   "reads_per_gpu": args.reads, "read_len": L_, "k": k, "tai": tai, "n_hash": nh, "batch_reads": args.batch_reads,
                   "sharding": "reads in file order; slice-wise prefix-OR(bloo1) + OR-allreduce(bloo2) over RCCL send/recv; walk handed rank to rank" if world > 1 else "single GPU"},
        "kmers_per_step": kmers_total, "lazy_flag_fallbacks": len(FALLBACKS),
        # what the ranks talked through: the size of the process group's communicator (backend nccl = RCCL), and every rank's HBM in use once the
        # steps are over (the library keeps its pools between steps: the device-wide figure is the run's high-water mark but for transient buffers)
        "rccl_ranks": (dist.get_world_size() if dist.is_initialized() and dist.get_backend() == "nccl" else 0),
        # set only by launch_ranks' second run: the run over RCCL failed or hung, this line is the same pipeline over gloo (NOT the xGMI number)
        "transport_fallback": os.environ.get("FAUCET_BENCH_FALLBACK_REASON"),
        "dist_backend": (dist.get_backend() if dist.is_initialized() else None), "hbm_per_rank": all_hbm,
        "outputs": {"junctions": int(sst["n_junctions"]) if world == 1 else None, "to_bloo2_rank0": int(lst["to_bloo2"]),
                    "walk_windows_rank0": int(sst["walk_windows"]), "walk_followers_rank0": int(sst["walk_followers"]),
                    "walk_max_cluster_rank0": int(sst["walk_max_cluster"]), "walk_key_ordered_pieces_rank0": int(sst["walk_parallel"]),
                    "flag_positions_rank0": int(sst["flag_positions"]), "piece_positions_rank0": int(sst["piece_positions"]),
                    "valid_reused_rank0": int(sst["valid_reused"]), "flags_filled_in_walk_rank0": int(sst["flags_filled"]), "nb_processed_rank0": int(sst["nb_processed"]),
                    "nb_skipped_rank0": int(sst["nb_skipped"]), "nb_jcheck_kmer_rank0": int(sst["nb_jcheck_kmer"])},
        "kernel_ms_per_step_rank0": {n: round(ms / max(prof_steps, 1), 3) for n, (c, ms) in sorted(ktimes.items(), key=lambda kv: -kv[1][1])},
        "ms_per_step_with_events": 1e3 * prof_elapsed / prof_steps if prof_steps else None,
        "timing_method": "value / ms_per_step: K steps WITHOUT HIP events around the kernels (since round 4; rounds 1-3 timed with them: +1.5 %); "
                         "ms_per_step_with_events: the bracketed steps behind them; N > 1 defaults to strong scaling on config 4 (rounds 1-3: weak)",
        "profiled_steps": {"steps": prof_steps, "ms_per_step": 1e3 * prof_elapsed / prof_steps if prof_steps else None,
                           "note": "separate steps of the same context with HIP events around every kernel, right behind the timed ones: where "
                                   "kernel_ms_per_step_rank0, roofline and device_time_share come from"},
    }
    if all_stage_ms is not None:
        res["rank_stage_ms"] = [None if st is None else [[n, round(ms, 3)] for n, ms in st] for st in all_stage_ms]
    if strong:
        merged = {}
        for c in all_checks or []:
            merged.update(c or {})
        res["outputs_check"] = merged or None
    # everything below that divides kernel times by steps / wall time refers to the bracketed steps
    steps_timed, elapsed_timed = args.steps, elapsed
    args.steps, elapsed = max(prof_steps, 1), (prof_elapsed if prof_steps else elapsed)

    # ---- roofline of the dominant kernel, from HIP events recorded on the context's stream inside the timed region
    T = None
    if not args.no_cpu:
        T = reference_bit_counts(k, L_, args.err, total_reads * L_ / total_genome, tai / E)
    heavy = {n: v for n, v in ktimes.items() if n in ("pack", "load_mark", "load_resolve", "scan_valid", "scan_flags")}
    if heavy:
        name = max(heavy, key=lambda n: heavy[n][1])
        launches, total_ms = heavy[name]
        kmers_per_launch = kmers_local * args.steps * (2 if name == "pack" else 1) / launches
        base_bytes = L_ / (L_ - k + 1)
        rho = lst["to_bloo2"] / max(kmers_local, 1)
        split = ctx.diag_load_split()                          # last load pass: occurrences routed to bloo2 by k_load_mark itself / left pending
        rho_mark = split["in_mark"] / max(kmers_local, 1)
        reused = sst["valid_reused"] / max(kmers_local, 1)     # validity answers taken from the load pass' planes: no filter access at all
        # ALGORITHMIC bytes per k-mer of each kernel = the 64-byte sectors its accesses NEED (never bytes that are not moved).
        # k_load_mark touches n_hash interleaved {bloo1, bloo2} words per k-mer: ONE sector each serves the test-and-set of bloo1 and the
        # set of bloo2, so it is charged 64 * n_hash -- not the 64 * n_hash * (1 + rho_mark) that the reference's two separate arrays
        # would move (kept beside it as `frac_reference_accesses`, with the counters' figure as `frac_measured_traffic`).  A validity answer
        # read from the resident `sure` plane moves no filter bytes (the reference's n_hash tests of those occurrences are charged to nobody).
        per_kmer = {
            "pack": base_bytes,                                  # each base read once per pass (1 B/base in HBM)
            "load_mark": 64.0 * nh,                              # one sector per interleaved word: test-and-set of bloo1 + bloo2's bit
            "load_resolve": 64.0 * nh * max(rho - rho_mark, 0.0),  # the bloo2 sets of the occurrences it settles
            "scan_valid": 64.0 * max((T["T_valid"] if T else nh) - nh * reused, 0.0),   # validity bit tests that still reach the filter
            "scan_flags": 64.0 * (T["T_junc"] if T else 0.0),    # alternate-extension + jcheck bit tests WITH the reference's skipping
        }[name]
        per_kmer_ref = {"pack": base_bytes, "load_mark": 64.0 * nh * (1.0 + rho_mark), "load_resolve": 64.0 * nh * max(rho - rho_mark, 0.0),
                        "scan_valid": 64.0 * (T["T_valid"] if T else nh), "scan_flags": 64.0 * (T["T_junc"] if T else 0.0)}[name]
        avg_ms = total_ms / launches
        achieved = per_kmer * kmers_per_launch / (avg_ms * 1e-3) / 1e9
        traffic = None          # HBM bytes per launch from rocprofv3 PMC passes, when a summary for this kernel is committed
        pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc_path):
            with open(pmc_path) as f:
                per_launch = {kk.split("<")[0]: vv for kk, vv in json.load(f).get("bytes_per_launch", {}).items()}
                traffic = per_launch.get("k_" + name, per_launch.get("k_" + name + "_sm"))   # scan_flags runs as k_scan_flags_sm for j <= 1
        res["device_time_share"] = {n: round(ms / (1e3 * elapsed), 4) for n, (c, ms) in ktimes.items() if ms / (1e3 * elapsed) > 0.01}
        res["roofline"] = {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                           "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "avg_launch_ms": avg_ms, "launches": launches,
                           "algorithmic_bytes_per_kmer": per_kmer, "kmers_per_launch": kmers_per_launch,
                           "attribution": "one 64-byte sector per access the kernel needs: n_hash interleaved {bloo1, bloo2} words per k-mer "
                                          "(the bloo2 sets ride in the same sector and are not charged again)",
                           # the committed counters' view of the same launches (FETCH_SIZE + WRITE_SIZE per launch / this run's mean duration)
                           "frac_measured_traffic": (traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if traffic else None,
                           # what the reference's two separate arrays would move for the accesses this kernel performs (round 2's headline)
                           "frac_reference_accesses": per_kmer_ref * kmers_per_launch / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                           "rho_settled_in_mark": rho_mark, "rho_settled_in_resolve": max(rho - rho_mark, 0.0)}
        # ---- what the counters say: FETCH_SIZE + WRITE_SIZE of every kernel of a step, from the committed PMC passes
        if os.path.exists(pmc_path):
            with open(pmc_path) as f:
                pl = {kk.split("<")[0]: vv for kk, vv in json.load(f).get("bytes_per_launch", {}).items()}
            alias = {"scan_flags": "k_scan_flags_sm", "load_resolve": "k_load_resolve_sm", "carry_update": "k_carry_from_first"}
            tot, missing = 0.0, []
            for n, (c, ms) in ktimes.items():
                b = pl.get("k_" + n, pl.get(alias.get(n, "")))
                if b is None:
                    if ms / (1e3 * elapsed) > 0.01 and n != "walk_stage":
                        missing.append(n)
                    continue
                tot += b * c / args.steps
            for n in ("k_walk_register", "k_walk_link", "k_walk", "k_walk_dyn", "k_walk_cluster", "k_walk_reset_uf"):   # the walk stage is timed as one entry
                if n in pl and "walk_stage" in ktimes:
                    tot += pl[n] * sst["walk_windows"]
            res["pipeline_measured"] = {"hbm_bytes_per_step": tot, "GBps": tot / (elapsed_timed / steps_timed) / 1e9, "frac_of_hbm_peak": tot / (elapsed_timed / steps_timed) / 1e9 / HBM_PEAK_GBPS,
                                        "source": "profiles/pmc_traffic.json (rocprofv3 FETCH_SIZE + WRITE_SIZE per launch, separate passes) x launches of this run",
                                        "kernels_without_counters": missing}
    if T:
        tsum = T["T_load"] + T["T_valid"] + T["T_junc"]
        ab64 = 2 * L_ / (L_ - k + 1) + 64.0 * tsum
        res["pipeline_ab64"] = {"bytes_per_kmer": ab64, "ab32_bytes_per_kmer": ab64 - 32.0 * tsum, "ab128_bytes_per_kmer": ab64 + 64.0 * tsum, "achieved_GBps": ab64 * value / 1e9, "frac_of_hbm_peak": ab64 * value / 1e9 / HBM_PEAK_GBPS,
                                "T_load": T["T_load"], "T_valid": T["T_valid"], "T_junc": T["T_junc"], "rho_sample": T["rho"],
                                "rho_gpu_run": lst["to_bloo2"] / max(kmers_local, 1),
                                # the device's own counters beside the sample's: validity probes it still sends to the filter, half-steps whose
                                # junction tests it evaluates (the preview's superset) against those the reference's walk tests
                                "T_valid_probed_gpu_run": max(T["T_valid"] - nh * sst["valid_reused"] / max(kmers_local, 1), 0.0),
                                "junction_test_halfsteps_gpu_run_per_kmer": 2.0 * sst["flag_positions"] / max(kmers_local, 1),
                                "junction_test_halfsteps_reference_per_kmer": sst["nb_processed"] / max(kmers_local, 1),
                                "counted_on": f"oracle, {T['sample_reads']} reads of a {T['sample_genome']} bp genome (same coverage, error rate, bits per estimated k-mer)"}

    # ---- measured ceilings of this device for the two access patterns of the path, ~1 s in total
    if not args.no_ceilings:
        n_acc = 1 << 28
        cl = {"stream_copy_GBps": ctx.diag_stream_copy(1 << 30, 5),
              "random_load32_per_s_filter": ctx.diag_random_access(tai // 8, n_acc, 0, 3),          # one load filter (probes)
              "random_test_or32_per_s_pair": ctx.diag_random_access(tai // 4, n_acc, 2, 3),         # both filters (Bloom::add)
              "random_atomic_min32_per_s_first": ctx.diag_random_access(tai * 4, n_acc, 1, 3),      # first-set times, 4 B per filter bit
              "filter_bytes": tai // 8, "first_bytes": tai * 4, "accesses_per_measurement": n_acc}
        cl["random_load_64B_sector_GBps"] = cl["random_load32_per_s_filter"] * 64 / 1e9
        props = torch.cuda.get_device_properties(device)
        cl["device"] = {"name": props.name, "hbm_bytes": props.total_memory, **ctx.diag_device_attr()}
        res["ceilings"] = cl
        if T:
            # random accesses the pipeline performs per second against what the device sustains for bare ones
            res["pipeline_ab64"]["bit_accesses_per_s"] = tsum * value
            res["pipeline_ab64"]["frac_of_random_access_ceiling"] = tsum * value / cl["random_load32_per_s_filter"]

    # ---- CPU baseline beside it (N = 1 only): the oracle on this host's cores, 1 thread like the reference
    if world == 1 and not args.no_cpu:
        n_s = min(args.cpu_sample_reads, args.reads)
        sample = reads[:n_s].cpu().numpy()
        v, dt, nk = cpu_baseline(sample, k, tai, nh)
        port = {"value": v, "unit": "k-mers/s", "cores": 1, "kind": "port",
                "sample": f"first {n_s} of the {args.reads} reads ({nk} k-mers), same 2 x {tai // 8 >> 20} MiB filters; "
                          f"load+scan took {dt:.1f} s on 1 of {os.cpu_count()} host cores"}
        ref = reference_binary_baseline(sample, k, E, S) if L_ - k + 1 > 0 else None
        if ref:     # the compiled reference itself is the stronger baseline; the port's number stays beside it
            rv, tl, ts = ref
            res["cpu_baseline"] = {"value": rv, "unit": "k-mers/s", "cores": 1, "kind": "reference",
                                   "sample": f"first {n_s} of the {args.reads} reads ({nk} k-mers) as FASTA through oracle/_ref/faucet_ref (the "
                                             f"reference's own sources, single-threaded), -estimated_kmers {E} -singletons {S}: load {tl:.1f} s + "
                                             f"scan {ts:.1f} s on 1 of {os.cpu_count()} host cores",
                                   "port": {"value": v, "seconds": dt}}
        else:
            res["cpu_baseline"] = port
        # the all-cores figure SURVEY 8d(ii) asks for beside the one-thread one: independent single-threaded replicas (see cpu_all_cores)
        workers = args.cpu_workers or min(os.cpu_count() or 1, 16)
        if workers > 1:
            ac = cpu_all_cores(sample, k, tai, nh, workers)
            if ac:
                res["cpu_baseline"]["all_cores"] = {"value": ac[0], "unit": "k-mers/s", "cores": workers, "kind": "port",
                                                    "sample": f"{workers} independent single-threaded replicas of the oracle, each load+scan over 1/{workers} of the "
                                                              f"same {n_s} reads with its own 2 x {tai // 8 >> 20} MiB filters ({ac[2]} k-mers in {ac[1]:.1f} s wall): the "
                                                              "work rate an ideal parallel port is bounded by, not the reference's result for the whole sample"}
    # ---- the PCIe-inclusive rate: the same step with the reads handed over as HOST buffers, 2 x 1 GB of copies inside
    # the timed region.  Reported beside `value`, never as `value`.
    if world == 1 and not args.host_input and not args.no_host_leg and not force_sharded and not strong:
        host = reads.cpu().numpy()
        hb = [api.ReadBatch.from_matrix(host[lo:hi]) for lo, hi in bounds]
        step_single(ctx, hb, pinned=True)
        ctx.synchronize()
        t0 = time.perf_counter()
        n_h = 3
        for _ in range(n_h):
            step_single(ctx, hb, pinned=True)
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / n_h
        res["host_input"] = {"value": kmers_local / dt, "unit": "k-mers/s", "ms_per_step": 1e3 * dt, "steps": n_h,
                             "note": "reads handed over as pageable host buffers: both passes copy them to the device inside the timed region"}
    # ---- Stage 3's walks on the step's own result: JunctionMap::findNeighbor from every junction along every covered
    # extension, whole walks on the device, host to host.  Beside `value`, never part of it.
    if world == 1 and not args.host_input and not args.no_host_leg and not force_sharded and not strong:
        try:
            starts, idx = [], []
            for i in range(5):
                m = (recs["dist"][:, i] > 0) & ((recs["cov"][:, i] > 0) if i < 4 else True)
                starts.append(np.asarray(keys)[m])
                idx.append(np.full(int(m.sum()), i, dtype=np.int8))
            starts, idx = np.concatenate(starts), np.concatenate(idx)
            t0 = time.perf_counter()
            ctx.stage3_set_junctions(np.asarray(keys), np.asarray(recs))
            t1 = time.perf_counter()
            nb, probes = ctx.stage3_find_neighbors(starts, idx, L_)
            t2 = time.perf_counter()
            res["stage3_find_neighbors"] = {"walks": int(len(starts)), "probes": int(probes), "seconds": t2 - t1, "set_map_seconds": t1 - t0,
                                            "value": probes / (t2 - t1), "unit": "getValidJExtension/s",
                                            "reached_a_junction": int((nb["node"] == 1).sum()), "asserts_tripped": int((nb["abort"] == 1).sum()),
                                            "note": "JunctionMap::findNeighbor for every (junction, covered extension) of the step's junction map in one "
                                                    "call, host arrays in, host arrays out"}
        except Exception as e:   # noqa: BLE001
            res["stage3_find_neighbors"] = {"error": repr(e)[:300]}
    # ---- file to files ("from the first byte of input consumed"): the `faucet` command line on the same reads as a FASTA file,
    # both passes reading it, `.bloom` and `.junctions` written.  Wall time of the whole process -- runtime start-up, context, output files
    # included -- beside `value`, never as `value`; the junction count must be the step's.
    if world == 1 and not args.host_input and not args.no_host_leg and not force_sharded and not strong:
        try:
            res["cli_file_to_files"] = cli_leg(reads, args, kmers_local, res["outputs"]["junctions"], variants=(("gpus2_one_device", ["-gpus", "2"]),))
        except Exception as e:   # noqa: BLE001  (a missing /tmp or binary must not cost the bench line)
            res["cli_file_to_files"] = {"error": repr(e)[:300]}
    # ---- the slowest configuration in the driver's line: BASELINE config 3's shape through the CLI, file to files
    if world == 1 and not args.host_input and not args.no_host_leg and not force_sharded and not strong:
        try:
            res["config3_cli"] = config3_cli_leg(device)
        except Exception as e:   # noqa: BLE001
            res["config3_cli"] = {"error": repr(e)[:300]}
    # ---- BASELINE's other configurations at full size, one cold step each (only config 2 had a driver-timed number)
    if world == 1 and not args.host_input and not args.no_host_leg and not force_sharded and not strong and not args.no_full_size:
        ctx.close()
        del reads, batches
        torch.cuda.empty_cache()
        res["full_size"] = {}
        for name, br in (("config5", 2_000_000), ("config4", 2_500_000)):
            try:
                res["full_size"][name] = full_size_leg(name, device, br)
            except Exception as e:   # noqa: BLE001
                res["full_size"][name] = {"error": repr(e)[:300]}
        # the same kernel on the LARGE filters, where it is furthest from the roofline: config 4's second step
        try:
            rl = res["full_size"]["config4"]["second_step"]["roofline_load_mark"]
            if rl:
                res["roofline_large"] = dict(rl, workload="BASELINE config 4 whole on one GPU (2 x 1 GiB filters), second step of the context")
        except (KeyError, TypeError):
            pass
    emit(json.dumps(res))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
