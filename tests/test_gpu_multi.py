"""`faucet -gpus N`: the C++ host that shards the reads over N GPUs from ONE process (faucet_amd/host/shard_host.h, faucet_amd/csrc/group.hip).

The test box has one GPU, so the N contexts share device 0 (the CLI says so on stderr) and the shards' bitmaps and tables travel by
device-to-device copies ordered with events -- the in-process transport; the RCCL transport is what a box with one device can show of it:
the library is loaded, a communicator made, bytes moved by ncclSend / ncclRecv (world size 1).  Every file is compared with the COMPILED
REFERENCE's golden (tests/golden/*, made by oracle/_ref/faucet_ref), dump order included."""
import ctypes as C
import gzip
import os
import subprocess

import numpy as np
import pytest

from tests.golden_util import Case

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "faucet_amd", "faucet")


def _run(c, tmp_path, gpus, extra=(), env=None, tag="out"):
    inp = str(tmp_path / ("reads.fq" if c.fastq else "reads.fa"))
    if not os.path.exists(inp):
        with open(inp, "wb") as f:
            f.write(c.reads_text())
    prefix = str(tmp_path / f"{tag}_{gpus}")
    # (round 5) from the third shard on a walk merges the keys created since its fresher preview into its planes: the library compares them with
    # planes made again, word by word, and says so on stderr if they ever differ
    r = subprocess.run([CLI, "-read_load_file", inp, "-read_scan_file", inp, "-file_prefix", prefix, "-gpus", str(gpus)] + c.meta["args"] + list(extra),
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, FGPU_DEBUG_DELTA_CHECK="1", **(env or {})))
    assert "merged in-map planes differ" not in r.stderr, r.stderr[-2000:]
    return prefix, r


def _same_files(c, prefix):
    for ext in ("bloom", "junctions", "short_pair_filter", "long_pair_filter"):
        gold = os.path.join(c.dir, f"out.{ext}.gz")
        if not os.path.exists(gold):
            assert not os.path.exists(prefix + "." + ext), ext
            continue
        with open(prefix + "." + ext, "rb") as f, gzip.open(gold, "rb") as g:
            assert f.read() == g.read(), "." + ext + " differs from the reference's"


@pytest.mark.parametrize("gpus", [2, 3, 4])
@pytest.mark.parametrize("case", ["c1_k21", "ragged_k31", "se_cleaning_k21", "mercy_k21", "pe_fastq_k21", "pe_repeats_k25", "pe_fasta_highcov_k31"])
def test_sharded_cli_writes_the_reference_files(case, gpus, tmp_path):
    """.bloom, .junctions (dump order included), .short_pair_filter and -- paired ends -- .long_pair_filter of a run sharded over 2, 3 and 4
    contexts are the compiled reference's, byte for byte; so are the counters it prints.  se_cleaning: the short pair filter handed from shard
    to shard; pe_*: the long pair filter (check-then-insert in file order) and the pair counts over the shards; mercy: the presence protocol."""
    c = Case(case)
    prefix, r = _run(c, tmp_path, gpus)
    assert r.returncode == (0 if c.no_cleaning else 3), r.stdout[-2000:] + r.stderr[-3000:]
    _same_files(c, prefix)
    cn = c.counters
    assert f"Distinct junctions: {cn['distinct_junctions']} " in r.stdout
    assert f"Number of processed kmers: {cn['nb_processed']} " in r.stdout
    assert f"Number of kmers that we j-checked: {cn['nb_jcheck_kmer']} " in r.stdout
    if c.paired:
        assert f"Empty count: {cn['empty_count']}, not empty count: {cn['not_empty_count']}" in r.stdout


@pytest.mark.parametrize("protocol", ["presence", "fixup", "fixup_planes"])
@pytest.mark.parametrize("extra", [[], ["-chunk_mb", "1"]], ids=["one_chunk", "chunks_1MB"])
def test_both_pass1_protocols_and_small_chunks(protocol, extra, tmp_path):
    """the fix-up protocol (own load, exclusive prefix-OR of bloo1, fgpu_load_fixup) and the presence protocol (presence bitmaps, prefix-OR,
    ordered load) both give the reference's filter; 1 MB chunks make every shard several batches"""
    for case in ("ragged_k31", "pe_repeats_k25"):
        c = Case(case)
        where = tmp_path / case
        where.mkdir()
        env = {"FAUCET_SHARD_PROTOCOL": protocol.split("_")[0], "FGPU_CLI_TIMES": "1"}
        if protocol == "fixup_planes":          # the fix-up's own pass with the fail planes (FGPU_LOAD_SHARD_PLANES: any shard size) instead of one clock
            env["FAUCET_SHARD_PLANES"] = "1"
        prefix, r = _run(c, where, 3, extra, env=env, tag=protocol)
        assert r.returncode == (0 if c.no_cleaning else 3), r.stdout[-2000:] + r.stderr[-3000:]
        assert ("fix-up protocol" if protocol.startswith("fixup") else "presence protocol") in r.stderr
        _same_files(c, prefix)


def test_a_rank_that_cannot_complete_its_pass_by_the_fixup_sends_all_ranks_to_the_presence_protocol(tmp_path):
    """a shard whose batches did not all stay resident (over the budget of fgpu_load_fixup_state, or no memory at that moment) left
    `faucet -gpus N` dead after a full pass 1.  The ranks now vote after their own loads; one "no" (forced here on rank 1) and every rank runs the
    presence protocol instead: one pass more, the reference's files."""
    for case in ("ragged_k31", "pe_repeats_k25"):
        c = Case(case)
        where = tmp_path / case
        where.mkdir()
        prefix, r = _run(c, where, 3, ["-chunk_mb", "1"], env={"FAUCET_DEBUG_FIXUP_NOT_READY": "1", "FGPU_CLI_TIMES": "1"}, tag="fallback")
        assert r.returncode == (0 if c.no_cleaning else 3), r.stdout[-2000:] + r.stderr[-3000:]
        assert "the presence protocol instead" in r.stderr and "pass 1 (shards, presence protocol)" in r.stderr
        _same_files(c, prefix)


def test_more_shards_than_records_and_refusals(tmp_path):
    """8 shards of a 30-record file leave shards empty; pipes and -batch_reads are refused loudly (a shard is a byte range of a regular file)"""
    c = Case("c1_k21")
    text = c.reads_text()
    few = b"\n".join(text.split(b"\n")[:60]) + b"\n"
    inp = tmp_path / "few.fa"
    inp.write_bytes(few)
    outs = {}
    for gpus in (1, 8):
        prefix = str(tmp_path / f"few_{gpus}")
        r = subprocess.run([CLI, "-read_load_file", str(inp), "-read_scan_file", str(inp), "-file_prefix", prefix, "-gpus", str(gpus)] + c.meta["args"],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        outs[gpus] = (open(prefix + ".bloom", "rb").read(), open(prefix + ".junctions", "rb").read())
    assert outs[1] == outs[8]
    r = subprocess.run([CLI, "-read_load_file", str(inp), "-read_scan_file", str(inp), "-file_prefix", str(tmp_path / "x"), "-gpus", "2", "-batch_reads", "100"] + c.meta["args"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 1 and "-batch_reads" in r.stderr
    fifo = str(tmp_path / "in.fifo")
    os.mkfifo(fifo)
    r = subprocess.run([CLI, "-read_load_file", fifo, "-read_scan_file", fifo, "-file_prefix", str(tmp_path / "y"), "-gpus", "2"] + c.meta["args"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 2 and "regular file" in r.stderr
    r = subprocess.run([CLI, "-read_load_file", str(inp), "-read_scan_file", str(inp), "-file_prefix", str(tmp_path / "z"), "-gpus", "2", "-transport", "rccl"] + c.meta["args"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 2 and "one device per rank" in r.stderr      # two shards on the box's one device: RCCL refuses, the CLI says why


@pytest.mark.parametrize("transport", [0, 1], ids=["copy", "rccl"])
def test_group_transports_move_bytes(transport):
    """the group's two transports on one device: a group of one rank is made, attached, and moves 1 MiB + 5 bytes from one buffer of the rank to
    another -- by event-ordered device copies, and by ncclSend / ncclRecv on a communicator of world size 1 (librccl loaded at run time)"""
    from faucet_amd import _lib as L
    from faucet_amd import api
    lib = L.load()
    ctx = api.Context(21, 1 << 20, 3)
    g = C.c_void_p()
    rc = lib.fgpu_group_create(1, transport, C.byref(g))
    assert rc == 0, lib.fgpu_group_last_error(g, -1)
    try:
        assert lib.fgpu_group_attach(g, 0, ctx.h) == 0, lib.fgpu_group_last_error(g, 0)
        ok = C.c_int(0)
        assert lib.fgpu_group_selftest(g, 0, (1 << 20) + 5, C.byref(ok)) == 0, lib.fgpu_group_last_error(g, 0)
        assert ok.value == 1
        # collectives of a world of one: the identity and zero
        ptr, nbytes = ctx.bloom_devptr(L.BLOO1)
        out = C.c_void_p()
        assert lib.fgpu_device_alloc(ctx.h, nbytes, C.byref(out)) == 0
        assert lib.fgpu_group_or_allreduce(g, 0, ptr, nbytes) == 0
        assert lib.fgpu_group_exclusive_prefix_or(g, 0, ptr, out, nbytes) == 0
        assert lib.fgpu_device_free(ctx.h, out) == 0
        assert lib.fgpu_group_barrier(g, 0) == 0
    finally:
        lib.fgpu_group_destroy(g)
        ctx.close()


def test_group_exchanges_between_four_contexts_on_one_device():
    """the slice exchanges themselves, four ranks (threads) on device 0, random bitmaps whose size the slices do not divide: OR-allreduce and
    exclusive prefix-OR against numpy"""
    import threading
    from faucet_amd import _lib as L
    from faucet_amd import api
    lib = L.load()
    n, nbytes = 4, (1 << 17)              # contexts with 2^20-bit filters: 128 KiB bitmaps; slices of 32 KiB
    rng = np.random.default_rng(11)
    for odd in (0, 48):                   # 48: a length the four slices do not divide evenly (16-byte granules)
        nb = nbytes - odd
        data = [rng.integers(0, 256, nb, dtype=np.uint8) & rng.integers(0, 256, nb, dtype=np.uint8) for _ in range(n)]
        ctxs = [api.Context(21, 1 << 20, 3) for _ in range(n)]
        g = C.c_void_p()
        assert lib.fgpu_group_create(n, L.TRANSPORT_COPY, C.byref(g)) == 0
        got_or, got_px, errs = [None] * n, [None] * n, []

        def rank(r):
            try:
                ctx = ctxs[r]
                assert lib.fgpu_group_attach(g, r, ctx.h) == 0
                ptr, _ = ctx.bloom_devptr(L.BLOO1)
                ctx.bloom_upload(L.BLOO1, np.concatenate([data[r], np.zeros(nbytes - nb, np.uint8)]))
                out = C.c_void_p()
                assert lib.fgpu_device_alloc(ctx.h, nbytes, C.byref(out)) == 0
                assert lib.fgpu_group_exclusive_prefix_or(g, r, ptr, out, nb) == 0, lib.fgpu_group_last_error(g, r)
                # the prefix is read back through bloo2's buffer
                p2, _ = ctx.bloom_devptr(L.BLOO2)
                assert lib.fgpu_device_zero(ctx.h, p2, nbytes) == 0
                assert lib.fgpu_device_copy(ctx.h, p2, out, nb) == 0
                got_px[r] = ctx.bloom_download(L.BLOO2)[:nb].copy()
                assert lib.fgpu_group_or_allreduce(g, r, ptr, nb) == 0, lib.fgpu_group_last_error(g, r)
                got_or[r] = ctx.bloom_download(L.BLOO1)[:nb].copy()
                assert lib.fgpu_device_free(ctx.h, out) == 0
            except BaseException as e:   # noqa: BLE001
                errs.append((r, e))
                lib.fgpu_group_abort(g)

        th = [threading.Thread(target=rank, args=(r,)) for r in range(n)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=120)
        lib.fgpu_group_destroy(g)
        for ctx in ctxs:
            ctx.close()
        assert not errs, errs
        want_or = data[0] | data[1] | data[2] | data[3]
        acc = np.zeros(nb, np.uint8)
        for r in range(n):
            assert np.array_equal(got_or[r], want_or), r
            assert np.array_equal(got_px[r], acc), r
            acc |= data[r]
