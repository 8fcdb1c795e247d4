"""Counter-based synthetic reads that come out IDENTICAL on a CPU and on a GPU.

`synth.py` (numpy) and `bench.py`'s generator (torch's device RNG) give data of the same shape but not the same bytes, so a
result computed here cannot be compared with one computed on the GPU box.  This generator uses only 64-bit integer
arithmetic on torch tensors (wrap-around multiply, xor, shifts): element i of a stream is a pure function of (seed, i),
whatever the device.  The full-size parity fixtures (tests/golden/fullsize.json, made by tests/golden/make_fullsize.py from the
oracle and from the compiled reference in the build container) are sha256 digests of the results on these reads; the GPU suite
regenerates the reads in HBM and compares digests.

Same model as SURVEY.md 8d: uniform random genome, reads placed uniformly, i.i.d. substitutions, half of the reads
reverse-complemented; paired reads = two ends of a fragment, mate 2 reverse-complemented, fragments flipped with p = 1/2.
"""
from __future__ import annotations

import torch

_M63 = (1 << 63) - 1


def _lsr(x, s):
    """logical shift right of an int64 tensor"""
    return (x >> s) & ((1 << (64 - s)) - 1)


def _c(v):
    """64-bit constant as a signed python int"""
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


def mix(seed: int, idx: torch.Tensor) -> torch.Tensor:
    """splitmix64 finaliser of (seed, idx); idx int64 tensor; result int64 (all 64 bits random)"""
    z = idx * _c(0x9E3779B97F4A7C15) + _c(seed * 0xD1342543DE82EF95 + 0x632BE59BD9B4E019)
    z = (z ^ _lsr(z, 30)) * _c(0xBF58476D1CE4E5B9)
    z = (z ^ _lsr(z, 27)) * _c(0x94D049BB133111EB)
    return z ^ _lsr(z, 31)


def _acgt(device):
    return torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)


def make_genome(length: int, seed: int, device, chunk: int = 1 << 26) -> torch.Tensor:
    out = torch.empty(length, dtype=torch.uint8, device=device)
    acgt = _acgt(device)
    for lo in range(0, length, chunk):
        n = min(chunk, length - lo)
        i = torch.arange(lo, lo + n, dtype=torch.int64, device=device)
        out[lo:lo + n] = acgt[_lsr(mix(seed, i), 62)]
    return out


def plant_repeats(genome: torch.Tensor, seed: int, families: int, copies: int, length: int, mut: float = 0.01) -> torch.Tensor:
    """Families of `copies` near-identical blocks of `length` bases (each copy with a few private substitutions) written over the
    genome in place: real branching for the junction scan, not only Bloom false positives.  Positions come from the same
    counter-based stream, drawn on the CPU (a handful of numbers), so the result does not depend on the device."""
    G = genome.numel()
    acgt = _acgt(genome.device)
    draws = (mix(seed, torch.arange(families * (copies + 1), dtype=torch.int64)) & _M63) % (G - length)
    draws = draws.tolist()
    for f in range(families):
        block = genome[draws[f * (copies + 1)]:draws[f * (copies + 1)] + length].clone()
        for c in range(copies):
            dst = draws[f * (copies + 1) + 1 + c]
            u = mix(seed + 1, torch.arange((f * copies + c) * length, (f * copies + c + 1) * length, dtype=torch.int64, device=genome.device))
            hit = _lsr(u, 40) < int(mut * (1 << 24))
            genome[dst:dst + length] = torch.where(hit, acgt[_lsr(u, 8) & 3], block)
    return genome


def _substitute(r, seed, first_row, err, device):
    """i.i.d. substitutions: element (row, col) of the read matrix draws from stream `seed` at index row * L + col"""
    n, L = r.shape
    acgt = _acgt(device)
    code = torch.zeros(256, dtype=torch.int64, device=device)
    code[acgt.long()] = torch.arange(4, device=device)
    idx = (torch.arange(first_row, first_row + n, dtype=torch.int64, device=device)[:, None] * L
           + torch.arange(L, dtype=torch.int64, device=device)[None, :])
    u = mix(seed, idx)
    hit = _lsr(u, 40) < int(err * (1 << 24))
    shift = (_lsr(u, 8) & 0xFFFF) % 3 + 1
    return torch.where(hit, acgt[(code[r.long()] + shift) & 3], r)


def _comp_table(device):
    comp = torch.zeros(256, dtype=torch.uint8, device=device)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    return comp


def make_reads(genome: torch.Tensor, n_reads: int, read_len: int, err: float, seed: int, device, chunk: int = 1_000_000,
               first_row: int = 0) -> torch.Tensor:
    """(n_reads, read_len) uint8 ASCII: rows first_row .. first_row + n_reads of the read set (a row is a pure function of
    (seed, row number), so a set can be made in slices — 200 M reads never have to exist in one place)"""
    G = genome.numel()
    comp = _comp_table(device)
    out = torch.empty((n_reads, read_len), dtype=torch.uint8, device=device)
    ar = torch.arange(read_len, dtype=torch.int64, device=device)
    for lo in range(0, n_reads, chunk):
        n = min(chunk, n_reads - lo)
        rid = torch.arange(first_row + lo, first_row + lo + n, dtype=torch.int64, device=device)
        u = mix(seed, rid)
        starts = (u & _M63) % (G - read_len + 1)
        r = genome[starts[:, None] + ar[None, :]]
        if err > 0:
            r = _substitute(r, seed + 1, first_row + lo, err, device)
        rc = (_lsr(mix(seed + 2, rid), 63) == 1)
        r = torch.where(rc[:, None], comp[r.long()].flip(1), r)
        out[lo:lo + n] = r
    return out


def make_pairs(genome: torch.Tensor, n_pairs: int, read_len: int, insert_lo: int, insert_hi: int, err: float, seed: int, device,
               chunk: int = 500_000) -> torch.Tensor:
    """Interleaved paired-end reads, (2 * n_pairs, read_len): row 2i and 2i+1 are the two ends of fragment i (insert size
    uniform in [insert_lo, insert_hi]), the second one reverse-complemented; fragments flipped (mates swapped) with p = 1/2."""
    G = genome.numel()
    comp = _comp_table(device)
    out = torch.empty((2 * n_pairs, read_len), dtype=torch.uint8, device=device)
    ar = torch.arange(read_len, dtype=torch.int64, device=device)
    for lo in range(0, n_pairs, chunk):
        n = min(chunk, n_pairs - lo)
        pid = torch.arange(lo, lo + n, dtype=torch.int64, device=device)
        ins = insert_lo + (mix(seed + 3, pid) & _M63) % (insert_hi - insert_lo + 1)
        starts = (mix(seed, pid) & _M63) % (G - insert_hi + 1)
        m1 = genome[starts[:, None] + ar[None, :]]
        m2 = comp[genome[(starts + ins - read_len)[:, None] + ar[None, :]].long()].flip(1)
        both = torch.empty((2 * n, read_len), dtype=torch.uint8, device=device)
        flip = (_lsr(mix(seed + 2, pid), 63) == 1)
        both[0::2] = torch.where(flip[:, None], m2, m1)
        both[1::2] = torch.where(flip[:, None], m1, m2)
        if err > 0:
            both = _substitute(both, seed + 1, 2 * lo, err, device)
        out[2 * lo:2 * (lo + n)] = both
    return out


def fasta_bytes(reads: torch.Tensor, fastq: bool = False) -> torch.Tensor:
    """The file text of the reads as a uint8 tensor on the reads' device: `>` + 9-digit index + newline + bases + newline
    (FASTQ: `@`..., then `+` and a constant quality line)."""
    n, L = reads.shape
    dev = reads.device
    w = 11 + L + 1 + ((2 + L + 1) if fastq else 0)
    rec = torch.empty((n, w), dtype=torch.uint8, device=dev)
    rec[:, 0] = ord("@") if fastq else ord(">")
    idx = torch.arange(n, dtype=torch.int64, device=dev)
    for d in range(9):
        rec[:, 9 - d] = (ord("0") + (idx // 10 ** d) % 10).to(torch.uint8)
    rec[:, 10] = ord("\n")
    rec[:, 11:11 + L] = reads
    rec[:, 11 + L] = ord("\n")
    if fastq:
        rec[:, 12 + L] = ord("+")
        rec[:, 13 + L] = ord("\n")
        rec[:, 14 + L:14 + 2 * L] = ord("I")
        rec[:, 14 + 2 * L] = ord("\n")
    return rec.reshape(-1)


def checksum(reads: torch.Tensor, first_row: int = 0) -> int:
    """Order-sensitive 64-bit checksum of a slice of a read matrix (rows first_row ..): sum over all bases of base * odd multiplier of its
    global index, in wrap-around int64 arithmetic -- the same number on a CPU and on a GPU, additive over row slices, and computed at memory
    speed where a sha256 of 20 GB of reads would take a minute of host time."""
    n, L = reads.shape
    dev = reads.device
    total = 0
    step = 1_000_000
    for lo in range(0, n, step):
        r = reads[lo:lo + step].to(torch.int64)
        idx = (torch.arange(first_row + lo, first_row + lo + r.shape[0], dtype=torch.int64, device=dev)[:, None] * L
               + torch.arange(L, dtype=torch.int64, device=dev)[None, :])
        total += int((r * (2 * mix(0x5EED, idx) + 1)).sum().item())
    return total & ((1 << 64) - 1)
