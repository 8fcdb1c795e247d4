#!/usr/bin/env python3
"""Golden of a SINGLE-END run with cleaning on (no --no_cleaning, no --paired_ends): the compiled reference writes `.bloom`, `.junctions` and
`.short_pair_filter` before its contig-graph stage starts (and may crash there: tolerated, the three files are complete by then).
    make -C oracle ref && python tests/golden/make_cleaning_golden.py
The product side fills that short pair filter on the device without bringing a single list to the host (fgpu_scan_short_pairs)."""
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402
from faucet_amd import synth  # noqa: E402

with tempfile.TemporaryDirectory() as td:
    g = synth.make_genome(6000, 41, repeats=3, repeat_len=150)
    r = synth.make_reads(g, 1500, 100, 0.01, 42)
    p = os.path.join(td, "se.fa")
    synth.write_fasta(p, r)
    G.run_case("se_cleaning_k21", p, False, ["-size_kmer", "21", "-max_read_length", "100", "-estimated_kmers", "100000", "-singletons", "20000"],
               tolerate_crash=True)
