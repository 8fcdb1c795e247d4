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
