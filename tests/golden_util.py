"""Helpers shared by the parity tests: load the committed golden fixtures (tests/golden/)."""
import gzip
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["c1_k21", "ragged_k31", "twohash_k31_L150", "j2_spacer20_k15", "j0_k15", "pe_fastq_k21", "mercy_k21", "nomercy_k21", "se_cleaning_k21",
         "pe_repeats_k25", "pe_fasta_highcov_k31", "pe_mercy_k21", "pe_twohash_k27"]      # (the last four: tests/golden/make_pairs_golden.py, round 4)


def _gz(path):
    with gzip.open(path, "rb") as f:
        return f.read()


def kat(kind):
    out = []
    with open(os.path.join(GOLDEN, "kat.jsonl")) as f:
        for line in f:
            d = json.loads(line)
            if d["kat"] == kind:
                out.append(d)
    return out


class Case:
    def __init__(self, name):
        self.name = os.path.basename(name)
        self.dir = name if os.path.isabs(name) else os.path.join(GOLDEN, name)     # (an absolute path: a case made on the spot, same layout)
        with open(os.path.join(self.dir, "case.json")) as f:
            self.meta = json.load(f)
        a = self.meta["args"]
        self.fastq = self.meta["fastq"]

        def opt(flag, default=None, cast=int):
            return cast(a[a.index(flag) + 1]) if flag in a else default

        self.k = opt("-size_kmer")
        self.E = opt("-estimated_kmers")
        self.S = opt("-singletons")
        self.j = opt("-j", 1)
        self.spacer = opt("-max_spacer_dist", 100)
        self.fp = opt("-fp", 0.04, float)
        self.max_read_length = opt("-max_read_length")
        self.paired = "--paired_ends" in a
        self.no_cleaning = "--no_cleaning" in a
        self.mercy = "--mercy" in a
        self.high_cov = "--high_cov" in a
        self.counters = self.meta["counters"]

    def pair_filter_elements(self):
        """(short, long) element counts the reference sizes its pair filters from (src/Faucet.cpp:266-283)"""
        E = self.E
        return (E // 2, E // 2) if self.high_cov else (E // 10, E // 5) if self.mercy else (E // 20, E // 10)

    def reads_text(self) -> bytes:
        return _gz(os.path.join(self.dir, "reads.fq.gz" if self.fastq else "reads.fa.gz"))

    def lines(self):
        """Sequence lines exactly as the reference's getline loop sees them
        (utils/Bloom.cpp:280-282,340): header, sequence, [plus, quality]."""
        txt = self.reads_text()
        raw = txt.split(b"\n")
        if raw and raw[-1] == b"":
            raw.pop()          # text ended with '\n': getline does not yield a final empty line
        step = 4 if self.fastq else 2
        out = []
        i = 0
        while i < len(raw):
            out.append(raw[i + 1] if i + 1 < len(raw) else b"")
            i += step
        return out

    def bloom(self) -> np.ndarray:
        return np.frombuffer(_gz(os.path.join(self.dir, "out.bloom.gz")), dtype=np.uint8)

    def junction_lines(self):
        return _gz(os.path.join(self.dir, "out.junctions.gz")).decode().split("\n")[:-1]

    def pair_filter(self, which) -> np.ndarray:
        return np.frombuffer(_gz(os.path.join(self.dir, f"out.{which}_pair_filter.gz")), dtype=np.uint8)
