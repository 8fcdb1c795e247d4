#!/usr/bin/env python3
"""Full-size parity fixtures: sha256 digests of what the ORACLE (oracle/liboracle.so) and the COMPILED REFERENCE
(oracle/_ref/faucet_ref) produce on BASELINE.json's configurations at their real sizes.

Run in the build container (hours of one CPU core in total; every case is independent and is merged into fullsize.json):

    python tests/golden/make_fullsize.py config2        # 10 M x 100 bp, k = 31, E = 1e8 / S = 2e7        oracle, ~6 min
    python tests/golden/make_fullsize.py config3        # 2.5 M pairs x 100 bp interleaved FASTQ, --paired_ends --fastq   reference binary, ~6 min
    python tests/golden/make_fullsize.py config5        # 50 M x 150 bp, 5 % errors, S/E = 0.5 -> 2 hashes   oracle, ~1.5 h
    python tests/golden/make_fullsize.py config2_cli    # config 2 as a FASTA file through the reference binary (.bloom / .junctions bytes)
    python tests/golden/make_fullsize.py config4        # 200 M x 100 bp, E = 1e9 / S = 2e8, 2^33-bit filters; streamed in slices   oracle, hours
    python tests/golden/make_fullsize.py checksum:config4   # (the reads' additive checksum, for an entry made before that existed)

The reads come from faucet_amd/synth_det.py (bit-identical on CPU and GPU); tests/test_gpu_fullsize.py regenerates them in HBM,
runs the device path and compares digests.  Stored: parameters, digest of the reads themselves (so that a generator mismatch is
told apart from a parity failure), digests of bloo1 / bloo2 / junction keys in creation order / junction records, counters.
"""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from faucet_amd import synth_det as sd  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.join(HERE, "fullsize.json")
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "faucet_ref")

CASES = {
    "config2": dict(genome=20_000_000, genome_seed=2, reads=10_000_000, read_len=100, err=0.01, read_seed=1000, k=31,
                    E=100_000_000, S=20_000_000),
    "config5": dict(genome=150_000_000, genome_seed=5, reads=50_000_000, read_len=150, err=0.05, read_seed=5000, k=31,
                    E=2_000_000_000, S=1_000_000_000),
    "config3": dict(genome=4_600_000, genome_seed=3, repeats=(40, 4, 500), pairs=2_500_000, read_len=100, insert=(250, 350), err=0.01,
                    read_seed=3000, k=31, E=100_000_000, S=20_000_000),
}
CASES["config2_cli"] = dict(CASES["config2"])
# BASELINE config 4 at its real size: 200 M x 100 bp of a 400 Mb genome, E = 1e9 / S = 2e8 -> 2^33-bit filters, 3 hashes.  Streamed: the
# reads are made `slice` rows at a time and handed to the oracle cumulatively (its filters and junction map carry over between calls),
# so 20 GB of reads never exist at once; checkpoint digests at every 1/8 of the reads = the shard boundaries of the 8-rank layout.
CASES["config4"] = dict(genome=400_000_000, genome_seed=4, reads=200_000_000, read_len=100, err=0.01, read_seed=4000, k=31,
                        E=1_000_000_000, S=200_000_000, slice=5_000_000, shards=8)


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def sha_file(path) -> str:
    h = hashlib.sha256()
    with open(path, "rb") as f:
        while True:
            b = f.read(1 << 24)
            if not b:
                break
            h.update(b)
    return h.hexdigest()


def make_case_reads(c, device="cpu"):
    g = sd.make_genome(c["genome"], c["genome_seed"], device)
    if "repeats" in c:
        sd.plant_repeats(g, c["genome_seed"] + 100, *c["repeats"])
    if "pairs" in c:
        return sd.make_pairs(g, c["pairs"], c["read_len"], c["insert"][0], c["insert"][1], c["err"], c["read_seed"], device)
    return sd.make_reads(g, c["reads"], c["read_len"], c["err"], c["read_seed"], device)


def merge(name, entry):
    data = {}
    if os.path.exists(OUT):
        with open(OUT) as f:
            data = json.load(f)
    data[name] = entry
    with open(OUT, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)
        f.write("\n")
    print(name, json.dumps(entry)[:600], flush=True)


def oracle_case(name):
    c = CASES[name]
    t0 = time.time()
    reads = make_case_reads(c).numpy()
    print(f"{name}: reads made in {time.time() - t0:.0f} s", flush=True)
    tai, nh, p1, bits = po.sizing_from_cli(c["E"], c["S"])
    bases, offs = po.reads_from_matrix(reads)
    entry = {"params": c, "tai": tai, "n_hash": nh, "reads_sha256": sha(reads), "made_by": "oracle/liboracle.so (pinned on the reference, tests/test_oracle_vs_golden.py)"}
    del reads
    t0 = time.time()
    b1, b2 = po.Bloom(tai, nh), po.Bloom(tai, nh)
    lst = po.load_two_filters(b1, b2, bases, offs, c["k"])
    entry.update(kmers=int(lst.kmers), to_bloo2=int(lst.to_bloo2), bloo1_sha256=sha(b1.bits()), bloo2_sha256=sha(b2.bits()),
                 load_seconds=round(time.time() - t0))
    print(f"{name}: load done in {time.time() - t0:.0f} s", flush=True)
    del b1
    t0 = time.time()
    sc = po.Scanner(c["k"], 1, 100, b2)
    sc.scan_reads(bases, offs)
    keys, recs = sc.junctions("creation")
    st = sc.stats()
    entry.update(keys_sha256=sha(keys), recs_sha256=sha(recs), dist_sha256=sha(recs["dist"]), cov_sha256=sha(recs["cov"]),
                 linked_sha256=sha(recs["linked"]), scan_seconds=round(time.time() - t0),
                 counters={k: int(st[k]) for k in ("n_junctions", "nb_jcheck_kmer", "nb_no_juncs", "nb_processed", "nb_skipped", "reads_no_errors",
                                                  "unambiguous_reads", "reads_processed")})
    merge(name, entry)


def oracle_case_streamed(name, limit_reads=None):
    """like oracle_case for a read set that is made and consumed in slices (config 4: 1.4e10 k-mers, hours of one core)"""
    c = dict(CASES[name])
    if limit_reads:                      # a cut-down rehearsal of the same procedure (not merged into the fixture file)
        c["reads"] = limit_reads
    g = sd.make_genome(c["genome"], c["genome_seed"], "cpu")
    tai, nh, p1, bits = po.sizing_from_cli(c["E"], c["S"])
    n, sl, shards = c["reads"], c["slice"], c["shards"]
    bounds = sorted({(n * (r + 1)) // shards for r in range(shards)})
    entry = {"params": c, "tai": tai, "n_hash": nh, "made_by": "oracle/liboracle.so (pinned on the reference, tests/test_oracle_vs_golden.py), reads streamed in slices"}

    def slices():
        lo = 0
        while lo < n:
            hi = min(n, lo + sl, min(b for b in bounds if b > lo))
            yield lo, hi, sd.make_reads(g, hi - lo, c["read_len"], c["err"], c["read_seed"], "cpu", first_row=lo).numpy()
            lo = hi

    t0 = time.time()
    b1, b2 = po.Bloom(tai, nh), po.Bloom(tai, nh)
    h = hashlib.sha256()
    kmers = to2 = cks = 0
    ck = []
    for lo, hi, reads in slices():
        h.update(np.ascontiguousarray(reads).tobytes())
        cks = (cks + sd.checksum(torch.from_numpy(reads), lo)) & ((1 << 64) - 1)
        bases, offs = po.reads_from_matrix(reads)
        lst = po.load_two_filters(b1, b2, bases, offs, c["k"])
        kmers += int(lst.kmers)
        to2 += int(lst.to_bloo2)
        if hi in bounds:
            ck.append({"reads": hi, "kmers": kmers, "to_bloo2": to2, "bloo1_sha256": sha(b1.bits()), "bloo2_sha256": sha(b2.bits())})
            print("CHECKPOINT load", json.dumps(ck[-1]), flush=True)
        print(f"{name}: load {hi} reads, {time.time() - t0:.0f} s", flush=True)
    entry.update(reads_sha256=h.hexdigest(), reads_checksum=cks, kmers=kmers, to_bloo2=to2, bloo1_sha256=ck[-1]["bloo1_sha256"], bloo2_sha256=ck[-1]["bloo2_sha256"],
                 load_checkpoints=ck, load_seconds=round(time.time() - t0))
    del b1
    t0 = time.time()
    sc = po.Scanner(c["k"], 1, 100, b2)
    ck = []
    names = ("n_junctions", "nb_jcheck_kmer", "nb_no_juncs", "nb_processed", "nb_skipped", "reads_no_errors", "unambiguous_reads", "reads_processed")
    for lo, hi, reads in slices():
        bases, offs = po.reads_from_matrix(reads)
        sc.scan_reads(bases, offs)
        if hi in bounds:
            keys, recs = sc.junctions("creation")
            st = sc.stats()
            ck.append({"reads": hi, "keys_sha256": sha(keys), "recs_sha256": sha(recs), "counters": {k: int(st[k]) for k in names}})
            print("CHECKPOINT scan", json.dumps(ck[-1]), flush=True)
        print(f"{name}: scan {hi} reads, {time.time() - t0:.0f} s", flush=True)
    keys, recs = sc.junctions("creation")
    st = sc.stats()
    entry.update(keys_sha256=sha(keys), recs_sha256=sha(recs), dist_sha256=sha(recs["dist"]), cov_sha256=sha(recs["cov"]),
                 linked_sha256=sha(recs["linked"]), scan_seconds=round(time.time() - t0), scan_checkpoints=ck,
                 counters={k: int(st[k]) for k in names})
    if limit_reads:
        print(json.dumps(entry)[:2000])
    else:
        merge(name, entry)


def reads_checksum_case(name):
    """the additive checksum of a streamed case's reads (synth_det.checksum: what the GPU test compares instead of a sha256 of 20 GB),
    added to an entry that was made before the checksum existed"""
    c = CASES[name]
    g = sd.make_genome(c["genome"], c["genome_seed"], "cpu")
    total = 0
    for lo in range(0, c["reads"], c["slice"]):
        n = min(c["slice"], c["reads"] - lo)
        total = (total + sd.checksum(sd.make_reads(g, n, c["read_len"], c["err"], c["read_seed"], "cpu", first_row=lo), lo)) & ((1 << 64) - 1)
        print(f"{name}: checksum over {lo + n} reads", flush=True)
    with open(OUT) as f:
        data = json.load(f)
    data[name]["reads_checksum"] = total
    with open(OUT, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)
        f.write("\n")
    print(name, "reads_checksum", total)


def reference_cli_case(name):
    """the reads as a file through the compiled reference itself; stopped once the hot path's files are written (its contig-graph
    stage is not part of the path and needs minutes to hours on these sizes)"""
    c = CASES[name]
    paired = "pairs" in c
    reads = make_case_reads(c)
    text = sd.fasta_bytes(reads, fastq=paired).numpy()
    entry = {"params": c, "reads_sha256": sha(reads.numpy()), "text_sha256": sha(text), "made_by": "oracle/_ref/faucet_ref (the reference's own sources)"}
    del reads
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        inp = os.path.join(td, "reads.fq" if paired else "reads.fa")
        text.tofile(inp)
        del text
        args = ["-size_kmer", str(c["k"]), "-max_read_length", str(c["read_len"]), "-estimated_kmers", str(c["E"]), "-singletons", str(c["S"])]
        args += ["--fastq", "--paired_ends"] if paired else ["--no_cleaning"]
        entry["args"] = args
        cmd = ["stdbuf", "-o0", REF_BIN, "-read_load_file", inp, "-read_scan_file", inp, "-file_prefix", os.path.join(td, "out")] + args
        t0 = time.time()
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        buf = b""
        while True:
            chunk = p.stdout.read1(65536)
            if not chunk:
                break
            buf += chunk
            if b"Number of junctions:" in buf:      # printed after all four files are closed (src/Faucet.cpp:296-306)
                break
        p.kill()
        p.wait()
        entry["seconds"] = round(time.time() - t0)
        out = buf.decode(errors="replace")
        for key, pat in (("distinct_junctions", r"Distinct junctions: (\d+)"), ("nb_jcheck_kmer", r"Number of kmers that we j-checked: (\d+)"),
                         ("nb_no_juncs", r"Number of reads with no junctions: (\d+)"), ("nb_processed", r"Number of processed kmers: (\d+)"),
                         ("nb_skipped", r"Number of skipped kmers: (\d+)"), ("reads_no_errors", r"Reads without errors: (\d+)"),
                         ("n_hash", r"Number of hash functions: (\d+)")):
            m = re.findall(pat, out)
            if m:
                entry[key] = int(m[0])
        w = re.findall(r"Weights after load: ([0-9.]+), ([0-9.]+)", out)
        if w:
            entry["weights_after_load"] = list(w[0])
        for ext in ("bloom", "junctions", "short_pair_filter", "long_pair_filter"):
            fp = os.path.join(td, "out." + ext)
            if os.path.exists(fp):
                entry[ext + "_sha256"] = sha_file(fp)
                entry[ext + "_bytes"] = os.path.getsize(fp)
    merge(name, entry)


if __name__ == "__main__":
    torch.set_num_threads(2)
    for name in sys.argv[1:]:
        if name.startswith("checksum:"):                  # "checksum:config4"
            reads_checksum_case(name.split(":")[1])
        elif name in ("config3", "config2_cli"):
            reference_cli_case(name)
        elif name.startswith("config4"):                 # "config4" or the rehearsal "config4:<reads>"
            oracle_case_streamed("config4", int(name.split(":")[1]) if ":" in name else None)
        else:
            oracle_case(name)
