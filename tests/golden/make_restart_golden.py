#!/usr/bin/env python3
"""Golden fixtures for the reference's two restart points (src/Faucet.cpp:97-109,130-134,185-195,257-258,289-293), made with the
COMPILED REFERENCE in the build container:  make -C oracle ref && python tests/golden/make_restart_golden.py

  restart_bloomfile_k21/           -bloom_file <a .bloom of a normal run>: the reference sizes the filter it loads into with
                                   create_bloom_filter_optimal(estimated_kmers, fpRate) -- fpRate, not p1 (src/Faucet.cpp:185-195) -- so it
                                   queries a 5-bit/3-hash file with 4 hash functions.  -estimated_kmers 80000 is chosen so that both sizings
                                   give 2^19 bits and the file has the size Bloom::load expects.  Stored: reads, the .bloom handed in, the
                                   .junctions the reference wrote from it, its counters.
  restart_bloomfile_twohash_k21/   the same with --two_hash: create_bloom_filter_2_hash(estimated_kmers, fpRate) = 10 bits, 2 hashes;
                                   -estimated_kmers 40000 gives 2^19 bits again (the .bloom comes from the 80000 run above).
  restart_junctions_k21/           -junctions_file <prefix> (with -bloom_file): the reference reloads <prefix>.junctions and the two pair filters
                                   (JunctionMap::buildFromFile, utils/JunctionMap.cpp:619-639) and goes straight to its contig graph; stored:
                                   the three files of a paired-end run and the "Number of junctions" line the reference prints after reloading.
"""
import gzip
import json
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from faucet_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "faucet_ref")
PATS = {"distinct_junctions": r"Distinct junctions: (\d+)", "nb_jcheck_kmer": r"Number of kmers that we j-checked: (\d+)",
        "nb_no_juncs": r"Number of reads with no junctions: (\d+)", "nb_processed": r"Number of processed kmers: (\d+)",
        "nb_skipped": r"Number of skipped kmers: (\d+)", "reads_no_errors": r"Reads without errors: (\d+)",
        "n_hash": r"Number of hash functions: (\d+)", "bits_per_kmer": r"Bits per kmer: (\d+)", "number_of_junctions": r"Number of junctions: (\d+)"}


def gz(path, data):
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        f.write(data)


def run(args, until=None):
    p = subprocess.Popen(["stdbuf", "-o0", REF] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    buf = b""
    while True:
        chunk = p.stdout.read1(65536)
        if not chunk:
            break
        buf += chunk
        if until and until in buf:
            break
    p.kill()
    p.wait()
    return buf.decode(errors="replace")


def counters(text):
    out = {}
    for k, pat in PATS.items():
        m = re.findall(pat, text)
        if m:
            out[k] = int(m[0])
    return out


def main():
    common = ["-size_kmer", "21", "-max_read_length", "100", "-singletons", "16000"]
    with tempfile.TemporaryDirectory() as td:
        g = synth.make_genome(4000, 51, repeats=3, repeat_len=150)
        r = synth.make_reads(g, 1000, 100, 0.01, 52)
        fa = os.path.join(td, "reads.fa")
        synth.write_fasta(fa, r)
        io = ["-read_load_file", fa, "-read_scan_file", fa]
        run(io + ["-file_prefix", os.path.join(td, "first"), "-estimated_kmers", "80000", "--no_cleaning"] + common)
        bloom = open(os.path.join(td, "first.bloom"), "rb").read()
        assert len(bloom) == (1 << 19) // 8
        for name, extra in (("restart_bloomfile_k21", ["-estimated_kmers", "80000"]),
                            ("restart_bloomfile_twohash_k21", ["-estimated_kmers", "40000", "--two_hash"])):
            out = os.path.join(HERE, name)
            os.makedirs(out, exist_ok=True)
            args = common + extra + ["--no_cleaning"]
            text = run(io + ["-file_prefix", os.path.join(td, name), "-bloom_file", os.path.join(td, "first.bloom")] + args)
            gz(os.path.join(out, "reads.fa.gz"), open(fa, "rb").read())
            gz(os.path.join(out, "in.bloom.gz"), bloom)
            gz(os.path.join(out, "out.junctions.gz"), open(os.path.join(td, name + ".junctions"), "rb").read())
            with open(os.path.join(out, "case.json"), "w") as f:
                json.dump({"name": name, "fastq": False, "args": args, "counters": counters(text)}, f, indent=1)
                f.write("\n")
            print(name, counters(text))
        # -junctions_file: a paired-end run's three files reloaded by the reference
        g = synth.make_genome(5000, 61, repeats=3, repeat_len=200)
        pr = synth.make_pairs(g, 500, 100, 300, 30, 0.01, 62)
        fq = os.path.join(td, "pe.fq")
        synth.write_fastq(fq, pr)
        pe_args = ["-size_kmer", "21", "-max_read_length", "100", "-estimated_kmers", "80000", "-singletons", "16000", "--fastq", "--paired_ends"]
        pre = os.path.join(td, "pe")
        run(["-read_load_file", fq, "-read_scan_file", fq, "-file_prefix", pre] + pe_args, until=b"Number of junctions:")
        text = run(["-read_load_file", fq, "-read_scan_file", fq, "-file_prefix", os.path.join(td, "again"), "-bloom_file", pre + ".bloom", "-junctions_file", pre]
                   + pe_args, until=b"Number of junctions:")
        out = os.path.join(HERE, "restart_junctions_k21")
        os.makedirs(out, exist_ok=True)
        for ext in ("bloom", "junctions", "short_pair_filter", "long_pair_filter"):
            gz(os.path.join(out, "in." + ext + ".gz"), open(pre + "." + ext, "rb").read())
        with open(os.path.join(out, "case.json"), "w") as f:
            json.dump({"name": "restart_junctions_k21", "fastq": True, "args": pe_args, "counters": counters(text),
                       "stdout_after_reload": [ln for ln in text.splitlines() if "junction" in ln.lower() or "pair filter" in ln.lower()]}, f, indent=1)
            f.write("\n")
        print("restart_junctions_k21", counters(text))


if __name__ == "__main__":
    main()
