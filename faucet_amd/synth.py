"""Seeded synthetic read generator (SURVEY.md §8d "Synthetic inputs").

Uniform random genome over ACGT; reads placed uniformly; each read reverse-complemented with
p = 0.5; i.i.d. substitution errors per base; optional N injection.  Output is either a
``(n_reads, read_len)`` uint8 matrix of ASCII bases or a FASTA / interleaved FASTQ file.

numpy only (this module must import on a CPU-only box); ``bench.py`` has a torch twin that
generates the same *shape* of data directly in HBM.
"""
from __future__ import annotations

import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTN", b"TGCAN"):
    _COMP[_a] = _b


def make_genome(length: int, seed: int, repeats: int = 0, repeat_len: int = 0) -> np.ndarray:
    rng = np.random.default_rng(seed)
    g = _ACGT[rng.integers(0, 4, size=length)]
    # optional planted repeats (copies of one random block) so that real junctions exist
    if repeats and repeat_len:
        src = int(rng.integers(0, length - repeat_len))
        block = g[src:src + repeat_len].copy()
        for _ in range(repeats):
            dst = int(rng.integers(0, length - repeat_len))
            g[dst:dst + repeat_len] = block
    return g


def make_reads(genome: np.ndarray, n_reads: int, read_len: int, err: float, seed: int,
               n_rate: float = 0.0) -> np.ndarray:
    """Return (n_reads, read_len) uint8 ASCII matrix."""
    rng = np.random.default_rng(seed)
    G = genome.shape[0]
    starts = rng.integers(0, G - read_len + 1, size=n_reads)
    idx = starts[:, None] + np.arange(read_len)[None, :]
    reads = genome[idx]
    # substitution errors: replace by one of the 3 other bases
    if err > 0:
        m = rng.random(reads.shape) < err
        code = np.zeros(256, dtype=np.uint8)
        code[_ACGT] = np.arange(4, dtype=np.uint8)
        c = code[reads]
        shift = rng.integers(1, 4, size=reads.shape).astype(np.uint8)
        reads = np.where(m, _ACGT[(c + shift) & 3], reads)
    # reverse-complement half of the reads
    rc = rng.random(n_reads) < 0.5
    reads[rc] = _COMP[reads[rc][:, ::-1]]
    if n_rate > 0:
        nm = rng.random(reads.shape) < n_rate
        reads = np.where(nm, np.uint8(ord("N")), reads)
    return np.ascontiguousarray(reads)


def write_fasta(path: str, reads: np.ndarray) -> None:
    n, L = reads.shape
    with open(path, "wb") as f:
        for i in range(n):
            f.write(b">r%d\n" % i)
            f.write(reads[i].tobytes())
            f.write(b"\n")


def write_fastq(path: str, reads: np.ndarray) -> None:
    """4-line records with a constant quality line (interleaved pairs = consecutive records)."""
    n, L = reads.shape
    q = b"I" * L
    with open(path, "wb") as f:
        for i in range(n):
            f.write(b"@r%d\n" % i)
            f.write(reads[i].tobytes())
            f.write(b"\n+\n")
            f.write(q)
            f.write(b"\n")


def make_pairs(genome: np.ndarray, n_pairs: int, read_len: int, insert_mean: int, insert_sd: int,
               err: float, seed: int) -> np.ndarray:
    """Interleaved paired-end reads: row 2i = mate 1 (forward strand), row 2i+1 = mate 2 (rc)."""
    rng = np.random.default_rng(seed)
    G = genome.shape[0]
    ins = np.clip(rng.normal(insert_mean, insert_sd, size=n_pairs).astype(np.int64), read_len, G)
    starts = rng.integers(0, G - ins + 1)
    ar = np.arange(read_len)[None, :]
    m1 = genome[starts[:, None] + ar]
    m2 = _COMP[genome[(starts + ins - read_len)[:, None] + ar][:, ::-1]]
    reads = np.empty((2 * n_pairs, read_len), dtype=np.uint8)
    reads[0::2] = m1
    reads[1::2] = m2
    if err > 0:
        m = rng.random(reads.shape) < err
        code = np.zeros(256, dtype=np.uint8)
        code[_ACGT] = np.arange(4, dtype=np.uint8)
        shift = rng.integers(1, 4, size=reads.shape).astype(np.uint8)
        reads = np.where(m, _ACGT[(code[reads] + shift) & 3], reads)
    # whole fragments flipped with p=0.5 (swap mates)
    flip = rng.random(n_pairs) < 0.5
    a = reads[0::2].copy()
    b = reads[1::2].copy()
    reads[0::2] = np.where(flip[:, None], b, a)
    reads[1::2] = np.where(flip[:, None], a, b)
    return np.ascontiguousarray(reads)
