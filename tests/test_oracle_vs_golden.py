"""Pin the oracle (oracle/faucet_oracle.cpp) against fixtures made by the compiled reference.

CPU only.  Fixtures: tests/golden/ (generator: tests/golden/make_golden.py, reference recipe:
oracle/Makefile `ref`).
"""
import os
import tempfile

import numpy as np
import pytest

from oracle import pyoracle as po
from tests.golden_util import CASES, Case, kat


def test_nt2int_kat():
    (d,) = kat("nt2int")
    got = [po.lib().fo_nt2int(c) for c in (b"A", b"C", b"T", b"G", b"N", b"a")]
    assert got == d["vals"]


@pytest.mark.parametrize("d", kat("hash"), ids=lambda d: f"k{d['k']}")
def test_codec_and_hash_kat(d):
    k, tai, read = d["k"], d["tai"], d["read"].encode()
    for p, e in enumerate(d["pos"]):
        f = po.lib().fo_encode(read[p:p + k], k)
        assert f == int(e["fwd"], 16)
        assert po.lib().fo_revcomp(f, k) == int(e["rc"], 16)
        c = po.lib().fo_canon(f, k)
        assert c == min(int(e["fwd"], 16), int(e["rc"], 16))
        assert po.lib().fo_old_hash(c, 0, tai) == e["hA"]
        assert po.lib().fo_old_hash(c, 1, tai) == e["hB"]


def test_seeds():
    # SURVEY.md A.2 (user_seed = 0)
    assert po.lib().fo_seed(0) == 0xffaa54ffe6e6e6e7
    assert po.lib().fo_seed(1) == 0x1140aada557088a4


def test_tai_kat():
    (d,) = kat("tai")
    for req, tai in d["cases"]:
        assert po.lib().fo_bloom_tai(req) == tai


@pytest.mark.parametrize("d", kat("sizing"), ids=lambda d: f"E{d['E']}_S{d['S']}_fp{d['fp']:.2f}")
def test_sizing_kat(d):
    p1, _ = po.solve_p1(d["E"], d["S"], d["fp"])
    assert p1 == d["p1"]          # bit-exact double: same solver, same operation order
    bits, tai, nh = po.size_optimal(d["E"], np.float32(p1))
    assert tai == d["tai"] and nh == d["n_hash"]


@pytest.mark.parametrize("d", kat("scan"), ids=lambda d: d["name"])
def test_readscan_kats(d):
    """The reference's own gtest inputs (src/newTests/ReadscanTest.cpp:102-264), answers taken
    from the reference's ReadScanner run on them (fake Bloom)."""
    CASE_INPUT = {
        "singleReadNoJunctions": (5, 0, 8, "ACGGG CGGGC GGGCG GGCGA GCGAA CGAAC GAACT AACTT ACTTT CTTTC TTTCA TTCAT TCATA CATAG ATAGG TAGGA",
                                  ["ACGGGCGAACTTTCATAGGA"]),
        "singleReadOneFakeJunction": (5, 0, 8, "ACGGG CGGGC GGGCG GGCGA GCGAA CGAAC GAACT AACTT AACTC ACTCC ACTTT CTTTC TTTCA TTCAT TCATA CATAG ATAGG TAGGA",
                                      ["ACGGGCGAACTTTCATAGGA"]),
        "LongReadNoJunctions": (5, 0, 8, "ACGGG CGGGC GGGCG GGCGA GCGAA CGAAC GAACT AACTT ACTTT CTTTC TTTCA TTCAT TCATA CATAG ATAGG TAGGA AGGAT GGATC GATCG ATCGC TCGCA CGCAC GCACT GCACT CACTC ACTCA CTCAC",
                                ["ACGGGCGAACTTTCATAGGATCGCACTCAC"]),
        "buildFullMap": (5, 0, 8, "ACGGG CGGGC GGGCG GGCGA GCGAA CGAAC GAACT AACTT ACTTT CTTTC TTTCA TTCAT TCATA CATAG ATAGG TAGGA GGCGA GCGAA CGAAC GAACT AACTA ACTAG CTAGT TAGTC AGTCC GTCCA TCCAT AACTT ACTTT CTTTC TTTCA TTCAT TCATA CATAC ATACG TACGA ACGAT CGATT",
                         ["ACGGGCGAACTTTCATAGGA", "GGCGAACTAGTCCAT", "AACTTTCATACGATT"]),
    }
    CASE_INPUT["buildFullMap_j1"] = (5, 1, 8) + CASE_INPUT["buildFullMap"][3:]
    CASE_INPUT["LongReadNoJunctions_j2_spacer4"] = (5, 2, 4, CASE_INPUT["LongReadNoJunctions"][3],
                                                    ["ACGGGCGAACTTTCATAGGATCGCACTCAC", "ACGGGCGAACTTTCANAGGATCGCACTCACNNACGGGCGAACT"])
    k, j, spacer, kmers, reads = CASE_INPUT[d["name"]]
    b = po.Bloom(1024, 2)
    b.fakify([po.lib().fo_canon(po.lib().fo_encode(s.encode(), k), k) for s in kmers.split()])
    sc = po.Scanner(k, j, spacer, b)
    for r in reads:
        sc.scan_input_read(r.encode())
    keys, recs = sc.junctions("map")
    got = po.junction_lines(keys, recs, k)
    assert got == d["junctions"]          # same std::unordered_map, same insertion order -> same iteration order


def check_oracle_against_case(c):
    """the oracle's load + scan on the case's reads against everything the compiled reference left of its run: counters it printed, .bloom,
    .junctions (byte order included), both pair filters"""

    bases, offs = po.reads_from_lines(c.lines())
    tai, nh, p1, bits = po.sizing_from_cli(c.E, c.S, c.fp)
    assert f"{p1:.6g}" == c.counters["p1"]
    assert bits == c.counters["bits_per_kmer"] and nh == c.counters["n_hash"]
    b1, b2 = po.Bloom(tai, nh), po.Bloom(tai, nh)
    st = po.load_two_filters(b1, b2, bases, offs, c.k, mercy=c.mercy)
    assert st.reads_processed == c.counters["load_reads_processed"]
    assert st.unambiguous_reads == c.counters["load_unambiguous"]
    assert np.array_equal(b2.bits(), c.bloom()), "bloo2 differs from the reference's .bloom"
    w = c.counters["weights_after_load"]
    assert f"{b1.weight():f}" == w[0] and f"{b2.weight():f}" == w[1]

    spf = lpf = None
    if not c.no_cleaning:
        n_short, n_long = c.pair_filter_elements()
        _, t_s, h_s = po.size_optimal(n_short, np.float32(0.01))
        spf = po.Bloom(t_s, h_s)
        if c.paired:
            _, t_l, h_l = po.size_optimal(n_long, np.float32(0.01))
            lpf = po.Bloom(t_l, h_l)
    sc = po.Scanner(c.k, c.j, c.spacer, b2, spf, lpf)
    sc.scan_reads(bases, offs, paired_ends=c.paired, no_cleaning=c.no_cleaning)
    s = sc.stats()
    cn = c.counters
    assert s["n_junctions"] == cn["distinct_junctions"]
    assert s["nb_jcheck_kmer"] == cn["nb_jcheck_kmer"]
    assert s["nb_no_juncs"] == cn["nb_no_juncs"]
    assert s["nb_processed"] == cn["nb_processed"]
    assert s["nb_skipped"] == cn["nb_skipped"]
    assert s["reads_no_errors"] == cn["reads_no_errors"]
    assert s["reads_processed"] == cn["scan_reads_processed"]
    assert s["unambiguous_reads"] == cn["scan_unambiguous"]
    assert s["empty_count"] == cn["empty_count"] and s["not_empty_count"] == cn["not_empty_count"]
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "o.junctions")
        sc.write_junctions(p)
        with open(p) as f:
            got = f.read().split("\n")[:-1]
    assert got == c.junction_lines(), ".junctions differs (byte order included)"
    if spf is not None:
        assert np.array_equal(spf.bits(), c.pair_filter("short"))
    if lpf is not None:
        assert np.array_equal(lpf.bits(), c.pair_filter("long"))


@pytest.mark.parametrize("name", CASES)
def test_end_to_end_vs_reference(name):
    check_oracle_against_case(Case(name))


def test_file_reader_matches_line_splitter(tmp_path):
    for name in ("ragged_k31", "pe_fastq_k21"):
        c = Case(name)
        p = tmp_path / ("r.fq" if c.fastq else "r.fa")
        p.write_bytes(c.reads_text())
        bases, offs = po.reads_from_file(str(p), c.fastq)
        b2, o2 = po.reads_from_lines(c.lines())
        assert np.array_equal(offs, o2) and np.array_equal(bases, b2)


@pytest.mark.parametrize("d", kat("stage3"), ids=lambda d: d["case"])
def test_stage3_probe_kats(d):
    """Stage 3's Bloom probes restated in the oracle against the reference's own JChecker::jcheck, JunctionMap::getValidJExtension
    and isBloomJunction on the golden filters"""
    c = Case(d["case"])
    b = po.Bloom(d["tai"], d["n_hash"])
    b.set_bits(c.bloom())
    lib = po.lib()
    for hx, contains, jc, ext, bj in d["probes"]:
        km = int(hx, 16)
        assert lib.fo_bloom_old_contains(b.h, lib.fo_canon(km, d["k"])) == contains
        assert lib.fo_stage3_jcheck(b.h, km, d["k"], d["j"]) == jc
        assert lib.fo_stage3_valid_extension(b.h, km, d["k"], d["j"]) == ext
        assert lib.fo_stage3_bloom_junction(b.h, km, d["k"], d["j"]) == bj
