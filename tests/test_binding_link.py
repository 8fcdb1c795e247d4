"""integration/faucet_binding.cpp LINKED into the compiled reference and RUN ("the drop-in is never linked").

`make -C oracle ref_stub` links the reference's own, unmodified objects (oracle/_ref/obj, compiled from /root/reference where it is mounted)
with the binding; the reference's two call sites of the hot path -- load_two_filters (src/Faucet.cpp:220) and ReadScanner::scanReads
(src/Faucet.cpp:244) -- are rerouted to it at link time (integration/wrap_shim.cpp, `-Wl,--wrap`).  Here the C ABI behind the binding is the
TEST stub (tests/stub/faucet_gpu_stub.cpp: the oracle on the CPU); tests/test_gpu_binding.py runs the same link against libfaucet_gpu.so on the
GPU box.  Everything after the two passes -- Bloom::dump, JunctionMap::writeToFile, the pair filters' dump, buildContigGraph, cleaning, the
contig files -- is the reference's own code running on what the ABI handed back: the run must end like the pure reference's (exit code,
every output file), which is what "the downstream ContigGraph stage is untouched" (north_star) means."""
import os
import re
import shutil
import subprocess

import pytest

from tests.golden_util import Case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
PURE = os.path.join(ROOT, "oracle", "_ref", "faucet_ref")

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "utils")) or shutil.which("g++") is None,
                                reason="the reference tree is not mounted here (the GPU box): see tests/test_gpu_binding.py")


@pytest.fixture(scope="module")
def linked():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref", "ref_stub"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return os.path.join(ROOT, "oracle", "_ref", "faucet_ref_stub")


def normalised(path):
    """file bytes with the node names of the .fastg (NODE_0x<heap address>_...) renumbered by first appearance: the reference prints pointers"""
    data = open(path, "rb").read()
    if not path.endswith(".fastg"):
        return data
    seen = {}
    return re.sub(rb"0x[0-9a-f]+", lambda m: seen.setdefault(m.group(0), b"n%d" % len(seen)), data)


def run_both(linked, case, tmp_path, extra=(), env_bound=None):
    c = Case(case)
    inp = str(tmp_path / ("reads.fq" if c.fastq else "reads.fa"))
    with open(inp, "wb") as f:
        f.write(c.reads_text())
    outs = {}
    for tag, exe in (("pure", PURE), ("bound", linked)):
        d = tmp_path / tag
        d.mkdir()
        r = subprocess.run([exe, "-read_load_file", inp, "-read_scan_file", inp, "-file_prefix", str(d / "out")] + c.meta["args"] + list(extra),
                           capture_output=True, text=True, timeout=600, env=dict(os.environ, **(env_bound or {})) if tag == "bound" else None)
        outs[tag] = (r, d)
    return c, outs


@pytest.mark.parametrize("case", ["se_cleaning_k21", "pe_fastq_k21", "c1_k21", "mercy_k21", "pe_repeats_k25", "pe_fasta_highcov_k31", "pe_mercy_k21", "pe_twohash_k27"])
def test_reference_with_the_binding_linked_in_ends_like_the_pure_reference(linked, case, tmp_path):
    c, outs = run_both(linked, case, tmp_path)
    (rp, dp), (rb, db) = outs["pure"], outs["bound"]
    assert rb.returncode == rp.returncode, (rp.returncode, rb.returncode, rb.stdout[-1500:], rb.stderr[-1500:])
    files = sorted(os.listdir(dp))
    assert files == sorted(os.listdir(db)) and any(f.endswith(".bloom") for f in files) and any(f.endswith(".junctions") for f in files)
    for f in files:                       # the hot path's files AND what the reference's own Stage 3 makes of them
        assert normalised(str(dp / f)) == normalised(str(db / f)), f
    if rp.returncode == 0 and "--no_cleaning" not in c.meta["args"]:
        assert any("contigs" in f for f in files), files          # the contig graph stage really ran to its end
    for line in ("Distinct junctions:", "Number of kmers that we j-checked:", "Number of processed kmers:", "Number of skipped kmers:"):
        want = [ln for ln in rp.stdout.splitlines() if ln.startswith(line)]
        got = [ln for ln in rb.stdout.splitlines() if ln.startswith(line)]
        assert want and want == got, line


@pytest.mark.parametrize("case", ["pe_fastq_k21", "pe_repeats_k25", "pe_fasta_highcov_k31"])
def test_the_bindings_own_paired_end_loop_ends_like_the_pure_reference(linked, case, tmp_path):
    """fgpu_scan_long_pairs answers FGPU_ERR_NOMEM (forced) -- the binding runs the reference's paired-end loop itself over the
    device's lists (host/pair_loop.h) into the reference's long_pair_filter, and the reference's Stage 3 ends as it does on its own passes"""
    c, outs = run_both(linked, case, tmp_path, env_bound={"FGPU_DEBUG_LONG_PAIRS_NOMEM": "1"})
    (rp, dp), (rb, db) = outs["pure"], outs["bound"]
    assert rb.returncode == rp.returncode, (rp.returncode, rb.returncode, rb.stdout[-1500:], rb.stderr[-1500:])
    assert "the paired-end loop runs on the host" in rb.stderr
    files = sorted(os.listdir(dp))
    assert files == sorted(os.listdir(db)) and any(f.endswith(".long_pair_filter") for f in files)
    for f in files:
        assert normalised(str(dp / f)) == normalised(str(db / f)), f
    counts = lambda out: [ln[ln.index("Empty count:"):] for ln in out.splitlines() if "Empty count:" in ln]      # noqa: E731
    assert counts(rp.stdout) and counts(rp.stdout) == counts(rb.stdout)


@pytest.mark.parametrize("gpus", [2, 3])
@pytest.mark.parametrize("case", ["se_cleaning_k21", "pe_repeats_k25", "mercy_k21"])
def test_the_linked_binding_over_read_shards_ends_like_the_pure_reference(linked, case, gpus, tmp_path):
    """the same link with FAUCET_GPUS=N: the binding runs both passes over N read shards (faucet_amd/host/shard_host.h: one host thread per
    shard, exchanges through the C ABI's group -- here the stub's), hands the reference its Bloom objects, junction map and pair filters, and
    the reference's own Stage 3 ends as it does on its own passes"""
    c, outs = run_both(linked, case, tmp_path, env_bound={"FAUCET_GPUS": str(gpus), "FAUCET_SHARD_PROTOCOL": "presence"})
    (rp, dp), (rb, db) = outs["pure"], outs["bound"]
    assert rb.returncode == rp.returncode, (rp.returncode, rb.returncode, rb.stdout[-1500:], rb.stderr[-1500:])
    files = sorted(os.listdir(dp))
    assert files == sorted(os.listdir(db))
    for f in files:
        assert normalised(str(dp / f)) == normalised(str(db / f)), f
