"""The PRODUCT against the COMPILED REFERENCE, no oracle in between (GPU box; the reference binary built in the build container travels with the
tree as test infrastructure): the `faucet` command line and oracle/_ref/faucet_ref run on the same random input -- the runs
of tests/test_oracle_vs_reference_fuzz.py::random_run -- and every file both write (.bloom, .junctions in dump order, .short_pair_filter,
.long_pair_filter) has to be the same bytes, and the scan's counters on stdout the same numbers."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "faucet_ref")
EXE = os.path.join(ROOT, "faucet_amd", "faucet")

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/faucet_ref was not built (make -C oracle ref)")]

LINES = ("Distinct junctions:", "Number of kmers that we j-checked:", "Number of reads with no junctions:", "Number of processed kmers:",
         "Number of skipped kmers:", "Reads without errors:", "Empty count:", "Reads processed:", "Unambiguous reads:", "Weights after load:")


def stdout_lines(text):
    """stdout without the progress lines and with the numbers of wall-clock lines blanked"""
    out = []
    for ln in text.replace("\r", "\n").splitlines():
        ln = ln.rstrip()
        if not ln or ln.startswith("reads consumed") or ln.startswith("reads scanned"):
            continue
        if re.search(r"Time|time|seconds", ln):
            ln = re.sub(r"[0-9.]+", "#", ln)
        out.append(ln)
    return out


@pytest.mark.parametrize("seed", range(100, 112))
def test_cli_equals_the_compiled_reference_on_a_random_run(seed, tmp_path, cli_extra=()):
    from tests.test_oracle_vs_reference_fuzz import random_run
    path, fastq, args = random_run(seed, tmp_path)
    outs = {}
    for tag, exe in (("ref", REF_BIN), ("gpu", EXE)):
        d = tmp_path / tag
        d.mkdir()
        # (unbuffered stdout: the reference may crash in its contig-graph stage, after the files and the counters this test is about)
        # (with -gpus N: the library checks the in-map planes a shard's walk MERGES with the new keys against planes made again, FGPU_DEBUG_DELTA_CHECK)
        r = subprocess.run(["stdbuf", "-o0", exe, "-read_load_file", path, "-read_scan_file", path, "-file_prefix", str(d / "out")] + args +
                           (list(cli_extra) if tag == "gpu" else []), capture_output=True, text=True, errors="replace", timeout=600,
                           env=dict(os.environ, FGPU_DEBUG_DELTA_CHECK="1") if tag == "gpu" and cli_extra else None)
        outs[tag] = (r, d)
    (rr, dr), (rg, dg) = outs["ref"], outs["gpu"]
    assert rg.returncode == (0 if "--no_cleaning" in args else 3), rg.stderr[-2000:]      # (the reference goes on into its contig graph and may crash there)
    assert "merged in-map planes differ" not in rg.stderr, rg.stderr[-2000:]
    mine = sorted(os.listdir(dg))
    assert any(f.endswith(".bloom") for f in mine) and any(f.endswith(".junctions") for f in mine)
    for f in mine:                                   # every file this build writes, the reference wrote too, with the same bytes
        assert os.path.exists(dr / f), f
        assert (dg / f).read_bytes() == (dr / f).read_bytes(), f
    for line in LINES:
        want = [ln.strip() for ln in rr.stdout.splitlines() if ln.startswith(line)]
        got = [ln.strip() for ln in rg.stdout.splitlines() if ln.startswith(line)]
        assert want == got, (line, want, got)
    assert re.search(r"Distinct junctions: \d+", rg.stdout)
    # ... and the log as a whole: what the command line prints is, line for line, what the reference prints up to the point where its
    # contig-graph stage begins (wall-clock numbers and the progress lines aside; the output prefix differs by construction)
    a = [ln.replace(str(dr), "<prefix>") for ln in stdout_lines(rr.stdout)]
    b = [ln.replace(str(dg), "<prefix>") for ln in stdout_lines(rg.stdout)]
    assert b and b[-1].startswith("Number of junctions:") and a[:len(b)] == b, [x for x in zip(a, b) if x[0] != x[1]][:5]


@pytest.mark.parametrize("gpus", [2, 3])
@pytest.mark.parametrize("seed", range(300, 306))
def test_cli_over_read_shards_equals_the_compiled_reference_on_a_random_run(seed, gpus, tmp_path):
    """the same with `-gpus N` (the C++ host over N read shards, faucet_amd/host/shard_host.h; all shards on the box's one device): random k, read
    length, FASTA / FASTQ, single / paired ends, cleaning, --mercy (presence protocol), reads with N, truncated reads, empty records -- the cuts fall
    wherever the bytes put them; every file and every line of the log as the compiled reference writes them"""
    test_cli_equals_the_compiled_reference_on_a_random_run(seed, tmp_path, cli_extra=["-gpus", str(gpus)])


def _same_files(da, db, crashed):
    out = []
    for fn in sorted(os.listdir(da)):
        a, b = os.path.join(da, fn), os.path.join(db, fn)
        if not os.path.exists(b):
            out.append((fn, "missing"))
            continue
        x, y = open(a, "rb").read(), open(b, "rb").read()
        if x != y and not (crashed and x[:len(y)] == y):     # (a reference that crashed in its contig graph may have left its last file in a buffer)
            out.append((fn, len(x), len(y)))
    return out


def restart_modes_differences(seed, tmp):
    """--just_load_bloom, then -bloom_file <that .bloom> for the scan (with and without --two_hash, which sizes the restarted filter: the file
    is usually NOT of that size, and Bloom::load, utils/Bloom.cpp:580-587, takes what fits), and --node_graph (a Stage-3 switch that must
    not change the hot path's files): [] if the command line wrote what the compiled reference wrote in every mode"""
    from tests.test_oracle_vs_reference_fuzz import random_run
    path, fastq, args = random_run(seed, tmp)
    args = [a for a in args if a != "--two_hash"]
    io = ["-read_load_file", path, "-read_scan_file", path]
    notes = []
    modes = ["just_load", "bloom_file", "bloom_file_two_hash", "node_graph"]
    if "--paired_ends" in args and "--no_cleaning" not in args:
        modes.append("junctions_file")       # (needs all three files of a run with cleaning; the reference loads both pair filters whatever the flags)
    for mode in modes:
        d = {tag: str(tmp / f"{mode}_{tag}") for tag in ("ref", "gpu")}
        for v in d.values():
            os.mkdir(v)
        if mode == "just_load":
            extra = ["--just_load_bloom"]
        elif mode.startswith("bloom_file"):
            extra = ["-bloom_file", str(tmp / "just_load_ref" / "out.bloom")] + (["--two_hash"] if mode.endswith("two_hash") else [])
        elif mode == "junctions_file":
            extra = ["-bloom_file", str(tmp / "just_load_ref" / "out.bloom"), "-junctions_file", str(tmp / "node_graph_ref" / "out")]
        else:
            extra = ["--node_graph"]
        r = {tag: subprocess.run(["stdbuf", "-o0", exe] + io + ["-file_prefix", os.path.join(d[tag], "out")] + args + extra,
                                 capture_output=True, text=True, errors="replace", timeout=600) for tag, exe in (("ref", REF_BIN), ("gpu", EXE))}
        diff = _same_files(d["gpu"], d["ref"], r["ref"].returncode < 0)
        a = [ln.replace(d["ref"], "<prefix>") for ln in stdout_lines(r["ref"].stdout)]
        b = [ln.replace(d["gpu"], "<prefix>") for ln in stdout_lines(r["gpu"].stdout)]
        if a[:len(b)] != b:                  # the log, line for line, up to where the reference's contig-graph stage begins
            diff.append(("stdout", [x for x in zip(a, b) if x[0] != x[1]][:3]))
        if diff or r["gpu"].returncode not in (0, 3) or (mode != "junctions_file" and not os.listdir(d["gpu"])):
            notes.append((mode, diff, r["gpu"].returncode, r["ref"].returncode, r["gpu"].stderr[-200:]))
    return notes


@pytest.mark.parametrize("seed", range(5000, 5006))
def test_restart_modes_equal_the_compiled_reference(seed, tmp_path):
    assert restart_modes_differences(seed, tmp_path) == []


BOUND = os.path.join(ROOT, "oracle", "_ref", "faucet_ref_gpu")


def binding_differences(seed, tmp):
    """the compiled reference with integration/faucet_binding.cpp linked in (its two hot-path call sites on libfaucet_gpu.so, everything else --
    the contig graph included -- the reference as compiled) against the pure reference on a random run: [] if exit status and every file,
    contig files included, are the same"""
    from tests.test_gpu_binding import normalised
    from tests.test_oracle_vs_reference_fuzz import random_run
    path, fastq, args = random_run(seed, tmp)
    res = {}
    for tag, exe in (("ref", REF_BIN), ("bound", BOUND)):
        (tmp / tag).mkdir()
        res[tag] = subprocess.run(["stdbuf", "-o0", exe, "-read_load_file", path, "-read_scan_file", path, "-file_prefix", str(tmp / tag / "out")] + args,
                                  capture_output=True, text=True, errors="replace", timeout=600)
    rr, rb = res["ref"], res["bound"]
    fr, fb = sorted(os.listdir(tmp / "ref")), sorted(os.listdir(tmp / "bound"))
    notes = []
    if rr.returncode != rb.returncode:
        notes.append(("exit", rr.returncode, rb.returncode, rb.stderr[-200:]))
    if fr != fb:
        notes.append(("files", fr, fb))
    for f in fb:
        if f in fr:
            x, y = normalised(str(tmp / "bound" / f)), normalised(str(tmp / "ref" / f))
            if x != y and not (rr.returncode < 0 and (x[:len(y)] == y or y[:len(x)] == x)):     # (a crash in the contig graph cuts the file being written)
                notes.append((f, len(x), len(y)))
    return notes, rr.returncode


@pytest.mark.skipif(not os.path.exists(BOUND), reason="oracle/_ref/faucet_ref_gpu was not built (make -C oracle ref_gpu)")
@pytest.mark.parametrize("seed", range(7000, 7006))
def test_linked_binding_ends_like_the_pure_reference_on_a_random_run(seed, tmp_path):
    notes, _ = binding_differences(seed, tmp_path)
    assert notes == []
