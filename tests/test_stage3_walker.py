"""faucet_amd/stage3.py (findNeighbor for many junctions in lock-step over the batched Stage-3 probes) against the reference's own
findNeighbor (tests/golden/stage3_neighbors_*.jsonl.gz, made by oracle/_ref/ref_kat neighbors).

CPU: the walker's host logic with the ORACLE answering getValidJExtension (checker only); GPU: the product path, fgpu_probe_valid_extension."""
import gzip
import json
import os

import numpy as np
import pytest

from faucet_amd import stage3
from oracle import pyoracle as po
from tests.golden_util import GOLDEN, Case

CASES = ["c1_k21", "ragged_k31", "twohash_k31_L150", "j2_spacer20_k15", "j0_k15"]
CODE = {"A": 0, "C": 1, "T": 2, "G": 3}


def _map_of(c):
    keys, recs = [], []
    for line in c.junction_lines():
        f = line.split()
        x = 0
        for ch in f[0]:
            x = (x << 2) | CODE[ch]
        keys.append(x)
        recs.append(([int(v) for v in f[6:10]], [int(v) for v in f[1:6]], [int(v) for v in f[11:16]]))
    return np.array(keys, dtype=np.uint64), np.array(recs, dtype=po.JUNC_DTYPE)


def _golden(name):
    with gzip.open(os.path.join(GOLDEN, f"stage3_neighbors_{name}.jsonl.gz"), "rt") as f:
        return [json.loads(line) for line in f if line.strip()]


def _check(name, ctx):
    c = Case(name)
    keys, recs = _map_of(c)
    kat = _golden(name)
    w = stage3.NeighborWalker(ctx, keys, recs, c.k, c.max_read_length)
    got = w.find_neighbors(np.array([int(e["start"], 16) for e in kat], dtype=np.uint64), np.array([e["index"] for e in kat]))
    for e, g in zip(kat, got):
        if e.get("abort"):
            assert g["abort"] == 1
            continue
        assert g["abort"] == 0
        assert (int(g["kmer"]), int(g["node"]), int(g["rindex"]), int(g["dist"]), int(g["len"])) == (int(e["kmer"], 16), e["node"], e["rindex"], e["dist"], e["len"]), e
    return w


class _OracleProbe:
    """stand-in for api.Context.probe_valid_extension (TEST ONLY): the oracle's restatement of getValidJExtension"""

    def __init__(self, c):
        self.k, self.j = c.k, c.j
        self.b = po.Bloom(len(c.bloom()) * 8, c.counters["n_hash"])
        self.b.set_bits(c.bloom())

    def probe_valid_extension(self, kmers):
        return np.array([po.lib().fo_stage3_valid_extension(self.b.h, int(x), self.k, self.j) for x in kmers], dtype=np.int8)


@pytest.mark.parametrize("name", CASES)
def test_walker_host_logic_against_the_reference_findNeighbor(name):
    _check(name, _OracleProbe(Case(name)))


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_walker_on_the_device_probes_against_the_reference_findNeighbor(name):
    from faucet_amd import _lib as L
    from faucet_amd import api
    c = Case(name)
    ctx = api.Context(c.k, len(c.bloom()) * 8, c.counters["n_hash"], j=c.j, max_spacer_dist=c.spacer)
    ctx.bloom_upload(L.BLOO2, c.bloom())
    w = _check(name, ctx)
    assert w.steps > 0 and w.probes >= w.steps       # one device call per lock-step round, many walks per call


def _check_device_walks(name, ctx):
    c = Case(name)
    keys, recs = _map_of(c)
    kat = _golden(name)
    ctx.stage3_set_junctions(keys, recs)
    starts, idx = np.array([int(e["start"], 16) for e in kat], dtype=np.uint64), np.array([e["index"] for e in kat])
    got, probes, contigs = ctx.stage3_find_neighbors(starts, idx, c.max_read_length, contigs=True)
    for e, g, text in zip(kat, got, contigs):
        if e.get("abort"):
            assert g["abort"] == 1
            continue
        assert g["abort"] == 0
        assert (int(g["kmer"]), int(g["node"]), int(g["rindex"]), int(g["dist"]), int(g["len"])) == (int(e["kmer"], 16), e["node"], e["rindex"], e["dist"], e["len"]), e
        assert text == e["contig"], e                       # BfSearchResult::contig, the sequence getContig strings together
    plain, probes2 = ctx.stage3_find_neighbors(starts, idx, c.max_read_length)
    assert probes2 == probes and plain.tobytes() == got.tobytes()
    return probes


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_whole_walks_on_the_device_against_the_reference_findNeighbor(name):
    """fgpu_stage3_find_neighbors: every probe, every k-mer step and every map look-up of a walk on the device, one lane per walk; equal to the
    reference's own findNeighbor on the golden maps, and to the lock-step walker probe for probe."""
    from faucet_amd import _lib as L
    from faucet_amd import api
    c = Case(name)
    ctx = api.Context(c.k, len(c.bloom()) * 8, c.counters["n_hash"], j=c.j, max_spacer_dist=c.spacer)
    ctx.bloom_upload(L.BLOO2, c.bloom())
    probes = _check_device_walks(name, ctx)
    w = _check(name, ctx)
    assert probes == w.probes


@pytest.mark.gpu
def test_whole_walks_on_the_device_equal_the_lock_step_walker_on_a_scanned_map():
    """a map two orders of magnitude larger than the goldens (the scan's own result on 60 k reads): every covered extension of every junction,
    device walks against the lock-step walker; plus the argument checks"""
    import bench
    import torch
    from faucet_amd import _lib as L
    from faucet_amd import api
    dev = torch.device("cuda", 0)
    n = 60_000
    reads = bench.make_reads(bench.make_genome(2 * n, 2, dev), n, 100, 0.01, 1000, dev)
    tai, nh = api.load_filter_shape(10 * n, 2 * n)
    ctx = api.Context(31, tai, nh)
    with pytest.raises(api.FaucetGpuError):
        ctx.stage3_find_neighbors(np.zeros(1, dtype=np.uint64), np.zeros(1, dtype=np.int8), 100)      # no map yet
    lst, sst, b2, keys, recs = bench.step_single(ctx, bench.device_batches(reads, 20_000))
    assert len(keys) > 2000
    starts, idx = [], []
    for i in range(5):
        m = (recs["dist"][:, i] > 0) & ((recs["cov"][:, i] > 0) if i < 4 else True)
        starts.append(keys[m])
        idx.append(np.full(int(m.sum()), i))
    starts, idx = np.concatenate(starts), np.concatenate(idx)
    ctx.stage3_set_junctions(keys, recs)
    got, probes = ctx.stage3_find_neighbors(starts, idx, 100)
    w = stage3.NeighborWalker(ctx, keys, recs, 31, 100)
    want = w.find_neighbors(starts, idx)
    for f in ("kmer", "node", "rindex", "dist", "len", "abort"):
        assert np.array_equal(got[f], want[f]), f
    assert probes == w.probes and (got["node"] == 1).sum() > 1000
    _, _, contigs = ctx.stage3_find_neighbors(starts[:5000], idx[:5000], 100, contigs=True)
    assert [len(t) for t in contigs] == [int(v) for v in got["len"][:5000]] and all(set(t) <= set("ACGT") for t in contigs)
    # a start that is no junction of the map, an index out of range: flagged per walk, the others are unaffected
    bad, _ = ctx.stage3_find_neighbors(np.array([starts[0], 12345, starts[1]], dtype=np.uint64), np.array([idx[0], 0, 7], dtype=np.int8), 100)
    assert list(bad["abort"]) == [int(got["abort"][0]), 2, 2] and bad["kmer"][0] == got["kmer"][0]
    with pytest.raises(api.FaucetGpuError):
        ctx.stage3_set_junctions(np.array([5, 5], dtype=np.uint64), recs[:2])                        # a k-mer twice
