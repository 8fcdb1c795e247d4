#!/usr/bin/env python3
"""Two more goldens of PAIRED-END runs with cleaning on, made by the compiled reference (round 4: both pair filters are built on the device now,
the long one as a fixed point of scanReads' check-then-insert loop, src/ReadScanner.cpp:317-343 -- more of the reference's own outputs to pin it):
    make -C oracle ref && python tests/golden/make_pairs_golden.py
  pe_repeats_k25        interleaved FASTQ of a small genome with eight planted repeats at high coverage (the same (k-mer, k-mer) pairs recur in
                        many read pairs: the order of checks and inserts decides the filter), with reads that hold N, truncated reads and
                        records whose sequence line is empty (each still toggles firstEnd: which reads are mates shifts behind them)
  pe_fasta_highcov_k31  interleaved FASTA (no --fastq), --high_cov (both pair filters sized from E / 2), k = 31, an odd number of records
  pe_mercy_k21          interleaved FASTQ with --mercy (pass 1 adds low-coverage runs to bloo2; pair filters sized from E / 10 and E / 5)
  pe_twohash_k27        interleaved FASTQ with --two_hash (the larger two-hash-function filters), reads of 150 bases
The reference may crash in its contig-graph stage (tolerated): the four files are complete by then."""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402
from faucet_amd import synth  # noqa: E402


def write_records(path, lines, fastq):
    with open(path, "wb") as f:
        for i, s in enumerate(lines):
            if fastq:
                f.write(b"@r%d\n" % i + s + b"\n+\n" + b"I" * len(s) + b"\n")
            else:
                f.write(b">r%d\n" % i + s + b"\n")


with tempfile.TemporaryDirectory() as td:
    rng = np.random.default_rng(4242)
    g = synth.make_genome(9000, 51, repeats=8, repeat_len=300)
    r = synth.make_pairs(g, 1203, 100, 250, 25, 0.01, 52)
    lines = [bytes(x) for x in np.ascontiguousarray(r)]
    for at in (7, 8, 901, 1500, 2200):                    # N inside a read: several pieces, the lists are spliced over them
        b = bytearray(lines[at]); b[int(rng.integers(20, 80))] = ord("N"); lines[at] = bytes(b)
    for at in (33, 1234):                                  # truncated reads (shorter than k: no piece at all)
        lines[at] = lines[at][:int(rng.integers(3, 24))]
    for at in (100, 101, 777, 2001):                       # empty records: the mates behind them pair up differently
        lines.insert(at, b"")
    p = os.path.join(td, "pe.fq")
    write_records(p, lines, True)
    G.run_case("pe_repeats_k25", p, True, ["-size_kmer", "25", "-max_read_length", "100", "-estimated_kmers", "150000", "-singletons", "30000",
                                           "--fastq", "--paired_ends"], tolerate_crash=True)

    g = synth.make_genome(7000, 61, repeats=4, repeat_len=220)
    r = synth.make_pairs(g, 700, 120, 330, 30, 0.008, 62)
    lines = [bytes(x) for x in np.ascontiguousarray(r)]
    lines.append(lines[5])                                  # an odd number of records: the last first end never meets a mate
    p = os.path.join(td, "pe.fa")
    write_records(p, lines, False)
    G.run_case("pe_fasta_highcov_k31", p, False, ["-size_kmer", "31", "-max_read_length", "120", "-estimated_kmers", "120000", "-singletons", "25000",
                                                  "--paired_ends", "--high_cov"], tolerate_crash=True)

    g = synth.make_genome(6000, 71, repeats=3, repeat_len=180)
    r = synth.make_pairs(g, 320, 100, 280, 25, 0.02, 72)              # 10x, 2 % errors: runs of low-coverage k-mers between solid ones
    p = os.path.join(td, "pem.fq")
    write_records(p, [bytes(x) for x in np.ascontiguousarray(r)], True)
    G.run_case("pe_mercy_k21", p, True, ["-size_kmer", "21", "-max_read_length", "100", "-estimated_kmers", "100000", "-singletons", "20000",
                                         "--fastq", "--paired_ends", "--mercy"], tolerate_crash=True)

    g = synth.make_genome(8000, 81, repeats=5, repeat_len=260)
    r = synth.make_pairs(g, 500, 150, 400, 40, 0.01, 82)
    p = os.path.join(td, "pe2.fq")
    write_records(p, [bytes(x) for x in np.ascontiguousarray(r)], True)
    G.run_case("pe_twohash_k27", p, True, ["-size_kmer", "27", "-max_read_length", "150", "-estimated_kmers", "200000", "-singletons", "100000",
                                           "--fastq", "--paired_ends", "--two_hash"], tolerate_crash=True)
