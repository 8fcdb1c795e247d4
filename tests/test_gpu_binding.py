"""The reference's OWN contig-graph stage on top of the GPU passes: oracle/_ref/faucet_ref_gpu is the compiled reference (unmodified
objects, built where /root/reference is mounted by `make -C oracle ref_gpu`) with integration/faucet_binding.cpp linked in and its two
call sites of the hot path rerouted to it (integration/wrap_shim.cpp) -- load_two_filters and ReadScanner::scanReads run on
libfaucet_gpu.so, everything else (sizing, Bloom::dump, JunctionMap::writeToFile, the pair filters, buildContigGraph, cleaning, the contig
files) is the reference as compiled.  The binary travels to the GPU box like the other built files under oracle/_ref; the expected files
are the PURE reference's, by digest (tests/golden/binding_stage3.json, made by tests/golden/make_binding_golden.py)."""
import hashlib
import json
import os
import re
import subprocess

import pytest

from tests.golden_util import Case

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "faucet_ref_gpu")
with open(os.path.join(ROOT, "tests", "golden", "binding_stage3.json")) as _f:
    WANT = json.load(_f)


def normalised(path):
    data = open(path, "rb").read()
    if not path.endswith(".fastg"):          # the reference names graph nodes by heap address
        return data
    seen = {}
    return re.sub(rb"0x[0-9a-f]+", lambda m: seen.setdefault(m.group(0), b"n%d" % len(seen)), data)


@pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref/faucet_ref_gpu was not built (it needs the reference tree: make -C oracle ref_gpu)")
@pytest.mark.parametrize("case", sorted(WANT))
def test_reference_stage3_runs_on_the_gpu_passes_and_ends_like_the_pure_reference(case, tmp_path):
    c = Case(case)
    inp = str(tmp_path / ("reads.fq" if c.fastq else "reads.fa"))
    with open(inp, "wb") as f:
        f.write(c.reads_text())
    r = subprocess.run([EXE, "-read_load_file", inp, "-read_scan_file", inp, "-file_prefix", str(tmp_path / "out")] + c.meta["args"],
                       capture_output=True, text=True, timeout=600)
    want = WANT[case]
    assert r.returncode == want["exit"], (r.returncode, r.stdout[-1500:], r.stderr[-1500:])
    got = {f: hashlib.sha256(normalised(str(tmp_path / f))).hexdigest() for f in sorted(os.listdir(tmp_path)) if f.startswith("out.")}
    assert sorted(got) == sorted(want["files"]), (sorted(got), sorted(want["files"]))
    for f, d in want["files"].items():       # .bloom / .junctions / pair filters AND the contig files the reference's Stage 3 writes
        assert got[f] == d, f
    for line in want["summary"]:
        assert line in r.stdout.splitlines(), line


@pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref/faucet_ref_gpu was not built (it needs the reference tree: make -C oracle ref_gpu)")
@pytest.mark.parametrize("case", [c for c in sorted(WANT) if c.startswith("pe_")])
def test_the_paired_end_loop_on_the_host_over_the_devices_lists(case, tmp_path):
    """VERDICT r5 item 6b on the device: the long pair filter's fixed-point form "does not fit" (FGPU_DEBUG_LONG_PAIRS_NOMEM) -- the binding runs the
    reference's loop over the lists the DEVICE hands out (fgpu_scan_take_stops) and every file, Stage 3's included, is the pure reference's"""
    c = Case(case)
    inp = str(tmp_path / ("reads.fq" if c.fastq else "reads.fa"))
    with open(inp, "wb") as f:
        f.write(c.reads_text())
    r = subprocess.run([EXE, "-read_load_file", inp, "-read_scan_file", inp, "-file_prefix", str(tmp_path / "out")] + c.meta["args"],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, FGPU_DEBUG_LONG_PAIRS_NOMEM="1"))
    want = WANT[case]
    assert r.returncode == want["exit"], (r.returncode, r.stdout[-1500:], r.stderr[-1500:])
    assert ("the paired-end loop runs on the host" in r.stderr) == ("--no_cleaning" not in c.meta["args"])
    got = {f: hashlib.sha256(normalised(str(tmp_path / f))).hexdigest() for f in sorted(os.listdir(tmp_path)) if f.startswith("out.")}
    for f, d in want["files"].items():
        assert got[f] == d, f
    for line in want["summary"]:
        assert line in r.stdout.splitlines(), line


@pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref/faucet_ref_gpu was not built (it needs the reference tree: make -C oracle ref_gpu)")
@pytest.mark.parametrize("gpus", [2, 3])
@pytest.mark.parametrize("case", sorted(WANT))
def test_reference_stage3_on_gpu_passes_over_read_shards(case, gpus, tmp_path):
    """FAUCET_GPUS=N: the linked binding runs both passes over N read shards (one host thread and one context per shard; on this box all on
    device 0) -- the reference's Stage 3 ends with the pure reference's files all the same"""
    c = Case(case)
    inp = str(tmp_path / ("reads.fq" if c.fastq else "reads.fa"))
    with open(inp, "wb") as f:
        f.write(c.reads_text())
    r = subprocess.run([EXE, "-read_load_file", inp, "-read_scan_file", inp, "-file_prefix", str(tmp_path / "out")] + c.meta["args"],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, FAUCET_GPUS=str(gpus)))
    want = WANT[case]
    assert r.returncode == want["exit"], (r.returncode, r.stdout[-1500:], r.stderr[-1500:])
    got = {f: hashlib.sha256(normalised(str(tmp_path / f))).hexdigest() for f in sorted(os.listdir(tmp_path)) if f.startswith("out.")}
    assert sorted(got) == sorted(want["files"]), (sorted(got), sorted(want["files"]))
    for f, d in want["files"].items():
        assert got[f] == d, f
