"""The oracle against the COMPILED REFERENCE on inputs made on the spot (CPU; where oracle/_ref/faucet_ref exists, i.e. where /root/reference was
mounted when `make -C oracle ref` ran).  The committed goldens pin the oracle on thirteen runs of the reference; this draws more: random k, read
length, error rate, FASTA / FASTQ, single / paired ends, cleaning on / off, --mercy, --high_cov, -j, -max_spacer_dist, reads with N, lower-case
letters, truncated reads and records with an empty sequence line -- each run of the reference is turned into a case of the goldens' layout in a
temporary directory and checked by the same function as the goldens (tests/test_oracle_vs_golden.py::check_oracle_against_case): counters the
reference printed, .bloom, .junctions in dump order, both pair filters."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "faucet_ref")

pytestmark = pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/faucet_ref is built where /root/reference is mounted (make -C oracle ref)")


def _write(path, lines, fastq):
    with open(path, "wb") as f:
        for i, s in enumerate(lines):
            if fastq:
                f.write(b"@r%d\n" % i + s + b"\n+\n" + b"I" * len(s) + b"\n")
            else:
                f.write(b">r%d\n" % i + s + b"\n")


def random_run(seed, tmp_path):
    """(path of the reads file, fastq, arguments) of a run drawn from `seed`"""
    from faucet_amd import synth
    rng = np.random.default_rng(9000 + seed)
    k = int(os.environ.get("FUZZ_K") or rng.choice([15, 21, 25, 31]))       # (FUZZ_K: scripts that draw other k, e.g. even ones)
    rl = int(rng.choice([60, 100, 150]))
    G_ = int(rng.integers(2000, 9000))
    g = synth.make_genome(G_, seed, repeats=int(rng.integers(0, 6)), repeat_len=int(min(G_ // 5, rng.integers(2 * k, 10 * k))))
    paired = bool(rng.integers(0, 2))
    err = float(rng.choice([0.0, 0.01, 0.03]))
    n = int(rng.integers(100, 900))
    if paired:
        r = synth.make_pairs(g, n, rl, int(rng.integers(rl, 3 * rl)), int(rng.integers(0, 30)), err, seed + 1)
    else:
        r = synth.make_reads(g, 2 * n, rl, err, seed + 1)
    lines = [bytes(x) for x in np.ascontiguousarray(r)]
    for _ in range(int(rng.integers(0, 8))):
        i = int(rng.integers(0, len(lines)))
        what = int(rng.integers(0, 4))
        if what == 0 and lines[i]:
            b = bytearray(lines[i]); b[int(rng.integers(0, len(b)))] = ord("N"); lines[i] = bytes(b)
        elif what == 1 and lines[i]:
            b = bytearray(lines[i]); j = int(rng.integers(0, len(b))); b[j] = ord(chr(b[j]).lower()); lines[i] = bytes(b)
        elif what == 2:
            lines[i] = lines[i][:int(rng.integers(0, len(lines[i]) + 1))]
        else:
            lines.insert(i, b"")
    fastq = bool(rng.integers(0, 2))
    E = int(rng.choice([60_000, 120_000, 200_000]))
    S = int(E // rng.choice([2, 5, 8]))
    args = ["-size_kmer", str(k), "-max_read_length", str(rl), "-estimated_kmers", str(E), "-singletons", str(S)]
    if fastq:
        args.append("--fastq")
    if paired:
        args.append("--paired_ends")
    if rng.integers(0, 3) == 0:
        args.append("--no_cleaning")
    if rng.integers(0, 5) == 0:
        args.append("--mercy")
    if rng.integers(0, 5) == 0:
        args.append("--high_cov")
    if rng.integers(0, 3) == 0:
        args += ["-j", str(int(rng.integers(0, 3)))]
    if rng.integers(0, 3) == 0:
        args += ["-max_spacer_dist", str(int(rng.choice([5, 20, 60])))]
    if os.environ.get("FUZZ_J"):                 # (scripts: larger j than the draws above take)
        args = [a for i, a in enumerate(args) if a != "-j" and (i == 0 or args[i - 1] != "-j")] + ["-j", os.environ["FUZZ_J"]]
    if os.environ.get("FUZZ_MORE_FLAGS"):        # (drawn behind everything else, so that the seeds of the suite keep their runs)
        if rng.integers(0, 3) == 0:
            args.append("--two_hash")            # sizes a filter restarted from a .bloom file; from reads it must not change anything
        if rng.integers(0, 3) == 0:
            args += ["-fp", str(float(rng.choice([0.02, 0.05, 0.1])))]
    path = str(tmp_path / ("in.fq" if fastq else "in.fa"))
    _write(path, lines, fastq)
    return path, fastq, args


@pytest.mark.parametrize("seed", range(16))
def test_oracle_equals_the_compiled_reference_on_a_random_run(seed, tmp_path):
    import make_golden as G
    from tests.golden_util import Case
    from tests.test_oracle_vs_golden import check_oracle_against_case
    path, fastq, args = random_run(seed, tmp_path)
    G.run_case("case", path, fastq, args, tolerate_crash=True, out_root=str(tmp_path))
    check_oracle_against_case(Case(str(tmp_path / "case")))
