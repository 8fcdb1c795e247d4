"""BASELINE.json's configurations at their FULL sizes on one MI355X, through size-independent properties (the oracle needs
minutes per configuration at these sizes, so it checks a sample; everything else is checked by invariants):

* determinism under re-batching: other batch boundaries, another walk-window span, eager flags, no resident planes and
  device-side record splitting must give the same bloo2 bytes and the same junction records in the same creation order;
* sharding: the two-shard protocol (presence bitmaps -> prefix-OR -> ordered load with kept carry -> OR of bloo2; table
  hand-over for the walk) gives the single-shard result;
* filter algebra: bloo2 is a subset of bloo1's bits (an occurrence only goes to bloo2 when all its bits are in bloo1), weight =
  popcount / tai, kmers = reads x (L - k + 1) for clean reads, to_bloo2 <= kmers;
* a prefix sample against the oracle: the load of the first reads alone (exactly what the oracle computes).
"""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest
import torch

import bench
from faucet_amd import _lib as L
from faucet_amd import api
from faucet_amd import synth_det as sd
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# digests of what the ORACLE / the COMPILED REFERENCE produce on the full-size configurations (tests/golden/make_fullsize.py, run
# once in the build container); the reads are regenerated here, on the device, by the same counter-based generator
with open(os.path.join(ROOT, "tests", "golden", "fullsize.json")) as _f:
    FULL = json.load(_f)


def _case_reads(name, dev):
    c = FULL[name]["params"]
    g = sd.make_genome(c["genome"], c["genome_seed"], dev)
    if "repeats" in c:
        sd.plant_repeats(g, c["genome_seed"] + 100, *c["repeats"])
    if "pairs" in c:
        r = sd.make_pairs(g, c["pairs"], c["read_len"], c["insert"][0], c["insert"][1], c["err"], c["read_seed"], dev)
    else:
        r = sd.make_reads(g, c["reads"], c["read_len"], c["err"], c["read_seed"], dev)
    return r


def _sha_file(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def _assert_equals_oracle_fixture(name, lst, sst, bloo1, bloo2, keys, recs):
    fx = FULL[name]
    assert lst["kmers"] == fx["kmers"] and lst["to_bloo2"] == fx["to_bloo2"]
    assert _digest(bloo1) == fx["bloo1_sha256"], "bloo1 differs from the oracle's at full size"
    assert _digest(bloo2) == fx["bloo2_sha256"], "bloo2 (the .bloom file) differs from the oracle's at full size"
    for key, want in fx["counters"].items():
        assert sst[key] == want, key
    assert _digest(keys) == fx["keys_sha256"], "junction keys / creation order differ from the oracle's at full size"
    assert _digest(recs["dist"]) == fx["dist_sha256"] and _digest(recs["cov"]) == fx["cov_sha256"] and _digest(recs["linked"]) == fx["linked_sha256"]
    assert _digest(recs) == fx["recs_sha256"]


def _digest(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _run(reads, k, tai, nh, batch_reads, **ctx_kw):
    ctx = api.Context(k, tai, nh, **ctx_kw)
    batches = bench.device_batches(reads, batch_reads)
    lst, sst, bloo2, keys, recs = bench.step_single(ctx, batches)
    bloo1 = ctx.bloom_download(L.BLOO1)
    w2 = ctx.bloom_weight(L.BLOO2)
    ctx.close()
    return lst, sst, bloo1, bloo2, keys, recs, w2


@pytest.fixture(scope="module")
def config2():
    """10 M x 100 bp, k = 31, estimated_kmers 1e8 / singletons 2e7 (BASELINE config 2), in HBM"""
    dev = torch.device("cuda", 0)
    reads = _case_reads("config2", dev)
    assert _digest(reads.cpu().numpy()) == FULL["config2"]["reads_sha256"], "the read generator gives other bytes here than in the build container"
    tai, nh = api.load_filter_shape(100_000_000, 20_000_000)
    base = _run(reads, 31, tai, nh, 1_000_000)
    return reads, tai, nh, base


def test_config2_full_size_equals_the_oracle(config2):
    """BASELINE config 2 at its FULL size, bit for bit against the oracle (pinned on the reference): bloo1 and bloo2 bytes, the
    973 k junction records in creation order, every counter -- by digest (the oracle needs 5 minutes of one core for this)."""
    reads, tai, nh, (lst, sst, bloo1, bloo2, keys, recs, w2) = config2
    assert (tai, nh) == (FULL["config2"]["tai"], FULL["config2"]["n_hash"])
    _assert_equals_oracle_fixture("config2", lst, sst, bloo1, bloo2, keys, recs)


def test_config2_filter_algebra_and_counters(config2):
    reads, tai, nh, (lst, sst, bloo1, bloo2, keys, recs, w2) = config2
    assert (tai, nh) == (1 << 29, 3)
    assert lst["kmers"] == 10_000_000 * 70 == sst["kmers"]
    assert 0 < lst["to_bloo2"] < lst["kmers"]
    assert not np.any(bloo2 & ~bloo1)                                   # bloo2's bits are a subset of bloo1's
    pop2 = int(np.unpackbits(bloo2).sum())
    assert f"{w2:f}" == f"{np.float32(pop2) / np.float32(tai):f}"
    assert sst["reads_processed"] == 10_000_000 and sst["n_junctions"] == len(keys) > 100_000
    assert len(np.unique(keys)) == len(keys)                            # one record per oriented junction k-mer
    assert sst["nb_processed"] + sst["nb_skipped"] > 0 and sst["valid_reused"] == lst["to_bloo2"]
    cov = recs["cov"].astype(np.int64).sum(axis=1)
    assert cov.min() >= 1                                               # every junction was visited by the read that created it


@pytest.mark.parametrize("variant", ["batches_2.5M", "span_2^18", "eager_flags", "no_resident", "batches_333333", "ramped_batches",
                                     "sweep_every_batch", "sweep_only_at_the_end"])
def test_config2_is_invariant_under_scheduling_choices(config2, variant, monkeypatch):
    reads, tai, nh, base = config2
    kw, batch = {}, 1_000_000
    if variant == "ramped_batches":                 # bench.py's default batching: small first and last batches
        batch = bench.batch_bounds(reads.shape[0], 1_000_000, 2)
    elif variant == "sweep_every_batch":            # the carry brought up to date after every batch / never before load_end
        monkeypatch.setenv("FGPU_SWEEP_RATIO", "0/1")
    elif variant == "sweep_only_at_the_end":
        monkeypatch.setenv("FGPU_SWEEP_RATIO", "1000000/1")
    if variant == "batches_2.5M":
        batch = 2_500_000
    elif variant == "batches_333333":
        batch = 333_333
    elif variant == "span_2^18":
        kw["walk_window_span"] = 1 << 18
    elif variant == "eager_flags":
        kw["eager_flags"] = True
    elif variant == "no_resident":
        kw["keep_resident"] = False
    lst, sst, bloo1, bloo2, keys, recs, _ = _run(reads, 31, tai, nh, batch, **kw)
    b = base
    assert lst["to_bloo2"] == b[0]["to_bloo2"]
    assert _digest(bloo2) == _digest(b[3]) and _digest(bloo1) == _digest(b[2])
    for key in ("n_junctions", "nb_jcheck_kmer", "nb_no_juncs", "nb_processed", "nb_skipped", "reads_no_errors", "unambiguous_reads"):
        assert sst[key] == b[1][key], key
    assert np.array_equal(keys, b[4])                                    # same records, same creation order
    assert _digest(recs) == _digest(b[5])


def test_config2_prefix_sample_against_the_oracle(config2):
    """the first 150 k reads alone: the device result must be the oracle's, bit for bit (same 64 MiB filters)"""
    reads, tai, nh, _ = config2
    n = 150_000
    sample = reads[:n].contiguous()
    lst, sst, bloo1, bloo2, keys, recs, _ = _run(sample, 31, tai, nh, 60_000)
    bases, offs = po.reads_from_matrix(sample.cpu().numpy())
    b1, b2 = po.Bloom(tai, nh), po.Bloom(tai, nh)
    olst = po.load_two_filters(b1, b2, bases, offs, 31)
    assert np.array_equal(bloo2, b2.bits()) and np.array_equal(bloo1, b1.bits())
    assert lst["to_bloo2"] == olst.to_bloo2
    osc = po.Scanner(31, 1, 100, b2)
    osc.scan_reads(bases, offs)
    okeys, orecs = osc.junctions("creation")
    assert np.array_equal(keys, okeys) and np.array_equal(recs["dist"], orecs["dist"]) and np.array_equal(recs["cov"], orecs["cov"])


def test_config2_two_shards_equal_one(config2):
    """reads 0..5M on shard A, 5M..10M on shard B (one process, two contexts): the exchange steps of DESIGN.md section 5"""
    reads, tai, nh, base = config2
    half = reads.shape[0] // 2
    parts = [bench.device_batches(reads[:half], 1_000_000), bench.device_batches(reads[half:], 1_000_000)]
    ctxs = [api.Context(31, tai, nh), api.Context(31, tai, nh)]
    pres = []
    for ctx, bs in zip(ctxs, parts):                   # presence bitmaps (would be all-gathered)
        ctx.load_begin(); ctx.load_end()
        for b in bs:
            ctx.presence_batch(b)
        pres.append(ctx.bloom_download(L.BLOO1))
    st = []
    for r, (ctx, bs) in enumerate(zip(ctxs, parts)):   # carried-in bloo1 = OR of the earlier shards' presence
        ctx.bloom_upload(L.BLOO1, np.zeros_like(pres[0]) if r == 0 else pres[0])
        ctx.load_begin(keep_carry=True)
        for b in bs:
            ctx.load_batch(b)
        st.append(ctx.load_end())
    bloo2 = ctxs[0].bloom_download(L.BLOO2) | ctxs[1].bloom_download(L.BLOO2)
    assert _digest(bloo2) == _digest(base[3])
    assert st[0]["to_bloo2"] + st[1]["to_bloo2"] == base[0]["to_bloo2"]
    for ctx in ctxs:
        ctx.bloom_upload(L.BLOO2, bloo2)
    # scan: both shards prepare, the walk goes A then B with the table handed over
    for ctx, bs in zip(ctxs, parts):
        ctx.scan_begin()
        for b in bs:
            ctx.scan_prepare(b)
    ctxs[0].scan_walk_prepared()
    sa = ctxs[0].scan_end()
    n_entries = ctxs[0].table_entries()
    table = torch.empty(max(n_entries, 1) * L.TABLE_ENTRY_BYTES, dtype=torch.uint8, device=reads.device)
    assert ctxs[0].export_table(table.data_ptr(), table.numel()) == n_entries
    ctxs[1].import_table(table.data_ptr(), n_entries, carried=sa)
    ctxs[1].scan_walk_prepared()
    sb = ctxs[1].scan_end()
    keys, recs = ctxs[1].junctions()
    assert np.array_equal(keys, base[4]) and _digest(recs) == _digest(base[5])
    for key in ("n_junctions", "nb_jcheck_kmer", "nb_no_juncs", "nb_processed", "nb_skipped", "reads_no_errors", "reads_processed"):
        assert sb[key] == base[1][key], key


def test_config5_two_hash_high_error_150bp():
    """BASELINE config 5's shape (--two_hash, 150 bp reads, 5 % errors: dense junctions) at 2 M reads: invariance under
    re-batching on the device, and a 60 k-read prefix against the oracle."""
    dev = torch.device("cuda", 0)
    k, Lr, n = 31, 150, 2_000_000
    reads = bench.make_reads(bench.make_genome(6_000_000, 5, dev), n, Lr, 0.05, 77, dev)
    bits, tai, nh = api.size_two_hash(400_000_000, 0.04)
    assert nh == 2
    a = _run(reads, k, tai, nh, 500_000)
    b = _run(reads, k, tai, nh, 777_777, walk_window_span=1 << 19)
    assert _digest(a[3]) == _digest(b[3]) and np.array_equal(a[4], b[4]) and _digest(a[5]) == _digest(b[5])
    assert a[0]["kmers"] == n * (Lr - k + 1)
    sample = reads[:60_000].contiguous()
    lst, sst, bloo1, bloo2, keys, recs, _ = _run(sample, k, tai, nh, 25_000)
    bases, offs = po.reads_from_matrix(sample.cpu().numpy())
    b1, b2 = po.Bloom(tai, nh), po.Bloom(tai, nh)
    po.load_two_filters(b1, b2, bases, offs, k)
    assert np.array_equal(bloo2, b2.bits())
    osc = po.Scanner(k, 1, 100, b2)
    osc.scan_reads(bases, offs)
    okeys, orecs = osc.junctions("creation")
    assert np.array_equal(keys, okeys) and np.array_equal(recs["dist"], orecs["dist"]) and np.array_equal(recs["linked"], orecs["linked"])


@pytest.mark.parametrize("ranks,per_rank,torch_stream,pairs", [(2, 2_000_000, False, False), (3, 700_000, False, False), (3, 700_000, True, False),
                                                               (3, 400_000, False, True), (2, 400_000, True, True)])
def test_ranks_in_separate_processes_sharing_this_gpu(ranks, per_rank, torch_stream, pairs):
    """bench.py's multi-GPU path (sharded.py over GpuShard) in real separate processes, one per rank, all on this GPU with gloo as
    the transport (scripts/two_rank_check.py): result identical to one context fed all shards in file order.  With the library on a
    stream of its own every exchange is fenced on the host (two fences were missing until late in round 2: one run in ten came out with a
    wrong bloo2 on three ranks); with the library on torch's stream nothing is."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GLOO_SOCKET_IFNAME="lo")      # one node: gloo must not look the host name up to find an interface
    if torch_stream:                                     # the library on torch's stream and no host fence, as bench.py runs N > 1
        env["FAUCET_TORCH_STREAM"] = "1"
    if pairs:                                            # both pair filters on, travelling with the table (round 5)
        env["FAUCET_CHECK_PAIRS"] = "1"
    r = None
    for attempt in range(2):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        try:
            r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1",
                                "--master-port", str(port), os.path.join(root, "scripts", "two_rank_check.py"), str(per_rank)],
                               capture_output=True, text=True, cwd=root, timeout=240, env=env)
        except subprocess.TimeoutExpired as e:           # (a run takes ten seconds) the rendezvous never came up
            r = subprocess.CompletedProcess(e.cmd, 124, (e.stdout or b"").decode(errors="replace") if isinstance(e.stdout, bytes) else (e.stdout or ""),
                                            "timed out after 240 s")
        if "RESULT" in r.stdout:          # the comparison was reached: its verdict stands, whatever it is
            break
        # Only a failed RENDEZVOUS is the harness' business and worth another port (the port picked above was taken in between, a rank did not
        # come up).  Once every rank has printed "INIT OK" (behind a barrier) the process group works: a run that then ends without a RESULT
        # -- a hang in the exchanges, a mismatched send/recv, a crash -- is the product's and fails here, with no second try to hide it.
        assert r.stdout.count("INIT OK") < ranks, "all ranks initialised, then no result (hang or crash in the sharded pipeline):\n" + \
            r.stdout[-3000:] + r.stderr[-3000:]
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "RESULT PASS" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def _bench_line(args, env, nproc=0, timeout=600, self_launch=0):
    """one run of bench.py (plain, or under torch.distributed.run with `nproc` ranks, or -- self_launch=N -- as the driver starts it: plain
    `python bench.py --gpus N`, which starts its own ranks as child processes); returns the parsed JSON line"""
    import socket
    import sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable]
    if nproc:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1", "--master-port", str(port)]
    cmd += [os.path.join(ROOT, "bench.py"), "--gpus", str(max(nproc, self_launch, 1))] + args
    base = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")} if self_launch else dict(os.environ)
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, timeout=timeout, env=dict(base, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), **env))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_sharded_path_over_rccl_gives_config2s_digests():
    """no driver-run test initialised the `nccl` backend.  bench.py's multi-GPU path (sharded.py: slice-wise prefix-OR / OR-allreduce,
    table hand-over, the library on torch's stream) with RCCL as the process group's backend at world size 1 (FAUCET_FORCE_SHARDED=1), strong
    mode on BASELINE config 2's fixture reads: bloo2, the junction keys in creation order, the records and the scan's counters equal the
    oracle's digests in tests/golden/fullsize.json."""
    d = _bench_line(["--scaling", "strong", "--fixture", "config2", "--batch-reads", "1000000", "--steps", "1", "--warmup", "0", "--no-cpu", "--no-ceilings",
                     "--no-host-leg", "--no-full-size", "--no-profile"], {"FAUCET_FORCE_SHARDED": "1"})
    assert d["n_gpus"] == 1 and d["scaling"] == "strong" and d["kmers_per_step"] == FULL["config2"]["kmers"]
    assert d["rccl_ranks"] == 1 and d["dist_backend"] == "nccl" and d["hbm_per_rank"][0]["used_bytes_after_the_steps"] > 0
    chk = d["outputs_check"]
    for key in ("bloo2_equals_the_oracles", "junction_keys_equal_the_oracles", "junction_records_equal_the_oracles", "scan_counters_equal_the_oracles"):
        assert chk[key] is True, (key, chk)
    stages = dict(d["rank_stage_ms"][0])
    assert "pass1_or_allreduce" in stages and "pass2_first_shard_scan" in stages


def test_bench_strong_scaling_three_ranks_equal_one():
    """`bench.py --gpus N --scaling strong` (the mode the 8-GPU target is measured in: ONE read set cut into N file-order shards) at reduced
    size: three ranks in separate processes sharing this GPU (gloo as the transport) give the digests of the one-GPU run of the same reads --
    bloo2, junction keys in creation order, records -- and report every rank's stage times."""
    size = ["--scaling", "strong", "--fixture", "config2", "--reads", "3000000", "--genome", "6000000", "--estimated-kmers", "30000000", "--singletons", "6000000",
            "--batch-reads", "250000", "--steps", "1", "--warmup", "1", "--no-cpu", "--no-ceilings", "--no-host-leg", "--no-full-size", "--no-profile"]
    one = _bench_line(size, {})
    # (round 6) started the way the driver starts N = 1: no launcher around it -- the parent spawns its ranks before any GPU call
    three = _bench_line(size, {"FAUCET_SHARE_GPU": "1", "FAUCET_DIST_BACKEND": "gloo", "GLOO_SOCKET_IFNAME": "lo"}, self_launch=3)
    assert three["n_gpus"] == 3 and three["rccl_ranks"] == 0 and three["dist_backend"] == "gloo" and len(three["hbm_per_rank"]) == 3
    assert one["scaling"] == three["scaling"] == "strong" and one["kmers_per_step"] == three["kmers_per_step"] == 3_000_000 * 70
    a, b = one["outputs_check"], three["outputs_check"]
    for key in ("bloo2_sha256", "junction_keys_sha256", "junction_records_sha256", "junctions"):
        assert a[key] == b[key], key
    assert len(three["rank_stage_ms"]) == 3 and all(st for st in three["rank_stage_ms"])
    assert "pass2_wait_for_table" in dict(three["rank_stage_ms"][2]) and "pass2_send" in dict(three["rank_stage_ms"][0])


def test_bench_falls_back_to_gloo_when_the_run_over_rccl_fails():
    """RCCL with N > 1 ranks has never run where this was built; `python bench.py --gpus N` must not lose a first contact with an N-GPU node to
    the transport.  Forced here: two ranks told to share this one GPU over `nccl` -- RCCL refuses two ranks on one device (or, at worst, never
    comes up: the parent's limit ends that) -- and the parent runs the same pipeline over gloo, says so in the line, and the outputs are the
    one-GPU run's."""
    size = ["--scaling", "strong", "--fixture", "config2", "--reads", "2000000", "--genome", "4000000", "--estimated-kmers", "20000000", "--singletons", "4000000",
            "--batch-reads", "250000", "--steps", "1", "--warmup", "0", "--no-cpu", "--no-ceilings", "--no-host-leg", "--no-full-size", "--no-profile"]
    one = _bench_line(size, {})
    two = _bench_line(size, {"FAUCET_SHARE_GPU": "1", "GLOO_SOCKET_IFNAME": "lo", "FAUCET_BENCH_RANKS_TIMEOUT": "120"}, self_launch=2)
    assert two["n_gpus"] == 2 and two["dist_backend"] == "gloo" and two["rccl_ranks"] == 0
    assert two["transport_fallback"] and "nccl" in two["transport_fallback"]
    assert one["transport_fallback"] is None
    for key in ("bloo2_sha256", "junction_keys_sha256", "junction_records_sha256", "junctions"):
        assert one["outputs_check"][key] == two["outputs_check"][key], key


@pytest.mark.skipif("config5" not in FULL, reason="tests/golden/fullsize.json has no config5 entry yet (make_fullsize.py config5: ~1.5 h of one core)")
def test_config5_full_size_equals_the_oracle():
    """BASELINE config 5 at its FULL size -- 50 M reads of 150 bases, 5 % errors, S/E = 0.5 so that the reference's own sizing gives two
    hash functions and 2 x 1 GiB filters -- bit for bit against the oracle, by digest."""
    dev = torch.device("cuda", 0)
    fx = FULL["config5"]
    c = fx["params"]
    reads = _case_reads("config5", dev)
    assert _digest(reads.cpu().numpy()) == fx["reads_sha256"]
    tai, nh = api.load_filter_shape(c["E"], c["S"])
    assert (tai, nh) == (fx["tai"], fx["n_hash"]) and nh == 2
    lst, sst, bloo1, bloo2, keys, recs, _ = _run(reads, c["k"], tai, nh, bench.batch_bounds(c["reads"], 2_000_000, 2))
    _assert_equals_oracle_fixture("config5", lst, sst, bloo1, bloo2, keys, recs)


def _run_cli(name, tmp_path, extra=()):
    """the reads of a full-size case as a FASTA / FASTQ file through the `faucet` command line; returns the output prefix"""
    dev = torch.device("cuda", 0)
    fx = FULL[name]
    reads = _case_reads(name, dev)
    assert _digest(reads.cpu().numpy()) == fx["reads_sha256"]
    paired = "pairs" in fx["params"]
    text = sd.fasta_bytes(reads, fastq=paired).cpu().numpy()
    del reads
    assert _digest(text) == fx["text_sha256"]
    inp = str(tmp_path / ("reads.fq" if paired else "reads.fa"))
    text.tofile(inp)
    del text
    prefix = str(tmp_path / "out")
    cli = os.path.join(ROOT, "faucet_amd", "faucet")
    r = subprocess.run([cli, "-read_load_file", inp, "-read_scan_file", inp, "-file_prefix", prefix] + fx["args"] + list(extra), capture_output=True, text=True, timeout=800)
    assert r.returncode == (0 if "--no_cleaning" in fx["args"] else 3), r.stdout[-2000:] + r.stderr[-2000:]
    os.remove(inp)
    return prefix, r.stdout


@pytest.mark.skipif("config3" not in FULL, reason="no config3 entry in tests/golden/fullsize.json")
def test_config3_paired_end_fastq_through_the_cli_equals_the_reference(tmp_path):
    """BASELINE config 3's shape at its real size: 2.5 M pairs of 100-base reads of a 4.6 Mb genome with planted repeats as an interleaved
    FASTQ file, `--fastq --paired_ends`.  All four files the hot path writes are byte-identical to the COMPILED REFERENCE's (by digest)."""
    fx = FULL["config3"]
    prefix, out = _run_cli("config3", tmp_path)
    for ext in ("bloom", "junctions", "short_pair_filter", "long_pair_filter"):
        assert os.path.getsize(prefix + "." + ext) == fx[ext + "_bytes"], ext
        assert _sha_file(prefix + "." + ext) == fx[ext + "_sha256"], "." + ext + " differs from the reference's"
    for label, key in (("Distinct junctions: ", "distinct_junctions"), ("Number of kmers that we j-checked: ", "nb_jcheck_kmer"),
                       ("Number of reads with no junctions: ", "nb_no_juncs"), ("Number of processed kmers: ", "nb_processed"),
                       ("Number of skipped kmers: ", "nb_skipped"), ("Reads without errors: ", "reads_no_errors")):
        assert f"{label}{fx[key]}" in out, label


@pytest.mark.skipif("config2_cli" not in FULL, reason="no config2_cli entry in tests/golden/fullsize.json")
def test_config2_fasta_through_the_cli_equals_the_reference(tmp_path):
    """config 2 as a 1.1 GB FASTA file, file to files: `.bloom` and `.junctions` (dump order included) equal the compiled reference's"""
    fx = FULL["config2_cli"]
    prefix, out = _run_cli("config2_cli", tmp_path)
    for ext in ("bloom", "junctions"):
        assert _sha_file(prefix + "." + ext) == fx[ext + "_sha256"], "." + ext + " differs from the reference's"


@pytest.mark.skipif("config2_cli" not in FULL, reason="no config2_cli entry in tests/golden/fullsize.json")
def test_config2_fasta_sharded_over_two_contexts_by_the_cli_equals_the_reference(tmp_path):
    """`faucet -gpus 2` on config 2's 1.1 GB FASTA file: the C++ host cuts the file into two file-order shards, one host thread and one context
    each (both on the box's one device), fix-up protocol in pass 1, the walk handed from shard to shard -- same bytes as the compiled reference"""
    fx = FULL["config2_cli"]
    prefix, out = _run_cli("config2_cli", tmp_path, ["-gpus", "2"])
    for ext in ("bloom", "junctions"):
        assert _sha_file(prefix + "." + ext) == fx[ext + "_sha256"], "." + ext + " differs from the reference's"


@pytest.mark.skipif("config3" not in FULL, reason="no config3 entry in tests/golden/fullsize.json")
def test_config3_paired_end_fastq_sharded_over_three_contexts_equals_the_reference(tmp_path):
    """config 3's shape with cleaning, `-gpus 3`: both pair filters travel with the junction table from shard to shard (the long one is
    check-then-insert in file order); all four files equal the compiled reference's"""
    fx = FULL["config3"]
    prefix, out = _run_cli("config3", tmp_path, ["-gpus", "3"])
    for ext in ("bloom", "junctions", "short_pair_filter", "long_pair_filter"):
        assert _sha_file(prefix + "." + ext) == fx[ext + "_sha256"], "." + ext + " differs from the reference's"
    for label, key in (("Distinct junctions: ", "distinct_junctions"), ("Number of processed kmers: ", "nb_processed"), ("Reads without errors: ", "reads_no_errors")):
        assert f"{label}{fx[key]}" in out, label


def test_config4_per_gpu_shape_is_invariant_under_scheduling_choices():
    """BASELINE config 4 on one of its eight GPUs: 25 M of the 200 M reads (a 6x shard of a 400 Mb genome) against filters sized for the whole
    run (-estimated_kmers 1e9 -singletons 2e8: 2 x 1 GiB, 32 GiB of first-set times).  No oracle at this size: the result must not depend on the
    batching, the walk's window span or the carry policy, and the filter algebra must hold."""
    dev = torch.device("cuda", 0)
    n = 25_000_000
    g = sd.make_genome(400_000_000, 4, dev)
    reads = sd.make_reads(g, n, 100, 0.01, 4000, dev)
    del g
    tai, nh = api.load_filter_shape(1_000_000_000, 200_000_000)
    assert (tai, nh) == (1 << 33, 3)
    a = _run(reads, 31, tai, nh, bench.batch_bounds(n, 2_500_000, 2))
    b = _run(reads, 31, tai, nh, 1_777_777, walk_window_span=1 << 22)
    assert a[0]["kmers"] == n * 70 and a[0]["to_bloo2"] == b[0]["to_bloo2"]
    assert _digest(a[3]) == _digest(b[3]) and _digest(a[2]) == _digest(b[2])        # bloo2, bloo1
    assert np.array_equal(a[4], b[4]) and _digest(a[5]) == _digest(b[5])            # junction records, creation order
    for key in ("n_junctions", "nb_jcheck_kmer", "nb_no_juncs", "nb_processed", "nb_skipped", "reads_no_errors"):
        assert a[1][key] == b[1][key], key
    assert not np.any(a[3] & ~a[2])                                                  # bloo2's bits are a subset of bloo1's


# ---- BASELINE config 4 at its FULL size: 200 M x 100 bp of a 400 Mb genome, -estimated_kmers 1e9 -singletons 2e8 (2 x 1 GiB filters, 3 hash
# functions), 1.4e10 k-mers -- on ONE MI355X (it fits: 20 GB of reads, 32 GiB of first-set times, the planes) and as the 8 file-order shards of
# the 8-GPU layout.  The oracle's digests (tests/golden/make_fullsize.py config4: hours of one core, streamed) include a checkpoint at every
# shard boundary, so the sharded run is checked after EVERY rank, not only at the end.
_partial = os.path.join(ROOT, "tests", "golden", "fullsize_partial.json")   # checkpoints of an oracle run still under way (development only)
if "config4" not in FULL and os.path.exists(_partial):
    with open(_partial) as _f:
        FULL["config4"] = json.load(_f)
needs_config4 = pytest.mark.skipif("config4" not in FULL, reason="tests/golden/fullsize.json has no config4 entry (make_fullsize.py config4: hours of one core)")


@pytest.fixture(scope="module")
def config4():
    dev = torch.device("cuda", 0)
    fx = FULL["config4"]
    c = fx["params"]
    free, total = torch.cuda.mem_get_info(dev)
    if total < 200 * (1 << 30):
        pytest.skip("config 4 at full size needs an MI355X-class device (288 GB)")
    g = sd.make_genome(c["genome"], c["genome_seed"], dev)
    reads = sd.make_reads(g, c["reads"], c["read_len"], c["err"], c["read_seed"], dev)
    del g
    # a generator mismatch is told apart from a parity failure: the additive checksum of the reads (synth_det.checksum, computed on the
    # device in a second) where the fixture has one, else the sha256 of all 20 GB through the host (~50 s)
    if "reads_checksum" in fx:
        total = 0
        for lo in range(0, c["reads"], 20_000_000):
            total = (total + sd.checksum(reads[lo:lo + 20_000_000], first_row=lo)) & ((1 << 64) - 1)
        assert total == fx["reads_checksum"] & ((1 << 64) - 1), "the read generator gives other bytes here than in the build container"
    elif "reads_sha256" in fx:
        h = hashlib.sha256()
        for lo in range(0, c["reads"], 10_000_000):
            h.update(reads[lo:lo + 10_000_000].cpu().numpy().tobytes())
        assert h.hexdigest() == fx["reads_sha256"], "the read generator gives other bytes here than in the build container"
    tai, nh = api.load_filter_shape(c["E"], c["S"])
    assert (tai, nh) == (fx["tai"], fx["n_hash"]) == (1 << 33, 3)
    yield reads, tai, nh, fx
    del reads
    torch.cuda.empty_cache()


def _sha_dev(t: torch.Tensor) -> str:
    """sha256 of a device byte tensor, copied out in pieces"""
    h = hashlib.sha256()
    for lo in range(0, t.numel(), 1 << 28):
        h.update(t[lo:lo + (1 << 28)].cpu().numpy().tobytes())
    return h.hexdigest()


@needs_config4
def test_config4_full_size_equals_the_oracle(config4):
    """BASELINE config 4, all 200 M reads through ONE context: bloo1, bloo2, the junction records in creation order and every counter equal
    the oracle's, by digest.  The first run of the path past 2^32 stream positions per pass (2.02e10) and with a junction table that grows
    many times."""
    reads, tai, nh, fx = config4
    if "bloo2_sha256" not in fx:
        pytest.skip("the oracle's final digests are not there yet")
    lst, sst, bloo1, bloo2, keys, recs, _ = _run(reads, fx["params"]["k"], tai, nh, bench.batch_bounds(reads.shape[0], 2_500_000, 2))
    assert not np.any(bloo2[:1 << 26] & ~bloo1[:1 << 26])
    _assert_equals_oracle_fixture("config4", lst, sst, bloo1, bloo2, keys, recs)


@needs_config4
@pytest.mark.parametrize("protocol", [pytest.param("presence", marks=pytest.mark.slow), "fixup", pytest.param("fixup_planes", marks=pytest.mark.slow)])
def test_config4_as_eight_shards_in_file_order_equals_the_oracle_after_every_rank(config4, protocol, monkeypatch):
    """The same reads as the 8 contiguous shards of the 8-GPU layout through the sharded pipeline's own steps (faucet_amd/sharded.py,
    run_in_turn: one process, the ranks' contexts made in turn so that 8 x 32 GiB of first-set times never coexist; the exclusive prefix-OR
    and the OR of bloo2 as slice-wise ORs on the device; the junction table handed from rank to rank, the hint from rank 0).  After EVERY
    rank the filters, the junction map and the counters are the sequential run's at that shard boundary = the oracle's checkpoint."""
    from faucet_amd import sharded
    if protocol == "presence":            # (round 5) in one of the three runs the library compares the planes it MERGED with the new keys against planes made again
        monkeypatch.setenv("FGPU_DEBUG_DELTA_CHECK", "1")
    if protocol == "fixup_planes":        # the fix-up's own pass with the fail planes (what shards beyond 2^32 positions take) instead of the shard-long clock
        monkeypatch.setenv("FAUCET_SHARD_PLANES", "1")
        protocol = "fixup"
    reads, tai, nh, fx = config4
    c = fx["params"]
    dev = reads.device
    world = c["shards"]
    n = reads.shape[0]
    cuts = [(n * r) // world for r in range(world + 1)]
    shards = [bench.device_batches(reads[cuts[r]:cuts[r + 1]], bench.batch_bounds(cuts[r + 1] - cuts[r], 2_500_000, 2)) for r in range(world)]
    load_ck = {ck["reads"]: ck for ck in fx.get("load_checkpoints", [])}
    scan_ck = {ck["reads"]: ck for ck in fx.get("scan_checkpoints", [])}
    checked = {"load": 0, "scan": 0}
    kmers = [0, 0]

    def after_load(r, stats, bloo1, bloo2):
        kmers[0] += stats["kmers"]
        kmers[1] += stats["to_bloo2"]
        ck = load_ck.get(cuts[r + 1])
        if ck is None:
            return
        assert (kmers[0], kmers[1]) == (ck["kmers"], ck["to_bloo2"]), f"k-mers / routed to bloo2 after shard {r}"
        assert _sha_dev(bloo1) == ck["bloo1_sha256"], f"bloo1 after shard {r} ({protocol})"
        assert _sha_dev(bloo2) == ck["bloo2_sha256"], f"bloo2 after shard {r} ({protocol})"
        checked["load"] += 1

    merged = []

    def after_scan(r, stats, backend):
        d = backend.ctx.diag_prepared_refresh()
        merged.append(d["batches_merged"])
        assert d["mismatching_words"] == 0, (r, d)
        ck = scan_ck.get(cuts[r + 1])
        if ck is None:
            return
        for key, want in ck["counters"].items():
            assert stats[key] == want, (r, key)
        keys, recs = backend.junctions()
        assert _digest(keys) == ck["keys_sha256"], f"junction keys / creation order after shard {r}"
        assert _digest(recs) == ck["recs_sha256"], f"junction records after shard {r}"
        checked["scan"] += 1

    lst, sst, last = sharded.run_in_turn(lambda: sharded.GpuShard(api.Context(c["k"], tai, nh), dev), shards, protocol, after_load, after_scan)
    last.close()
    assert merged[0] == 0 and all(m > 0 for m in merged[2:]), merged      # from the third rank on the walk merges the new keys into the planes (a fresher preview came first)
    assert checked["load"] >= min(1, len(load_ck)) and checked["scan"] >= min(1, len(scan_ck))
    if "bloo2_sha256" in fx:         # the whole fixture is there: every shard boundary was compared
        assert checked == {"load": world, "scan": world}
