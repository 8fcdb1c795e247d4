"""Degenerate inputs through the command line against the compiled reference (GPU box): empty file, reads shorter than k, exactly k, only N, no
final newline, CRLF line ends, blank lines, a FASTQ whose last record is cut short, one very short and one long read.  Every file both write must
be the same bytes (the reference may crash after writing: tolerated)."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF, EXE = os.path.join(ROOT, "oracle", "_ref", "faucet_ref"), os.path.join(ROOT, "faucet_amd", "faucet")
A = b"ACGTTGCATGCCGATTACGGATCCATGCAAGTCGATCGGATTCAGCATGCATCGATCGGGCTAAGCTAGCTTTAGCGATCGA"
CASES = {
    "empty file": (b"", False),
    "one read shorter than k": (b">r\nACGTACGT\n", False),
    "one read of exactly k": (b">r\n" + A[:21] + b"\n", False),
    "only N": (b">r\nNNNNNNNNNNNNNNNNNNNNNNNNNNNNNN\n>s\nNNNN\n", False),
    "no final newline": (b">r\n" + A + b"\n>s\n" + A[5:60], False),
    "CRLF line ends": (b">r\r\n" + A + b"\r\n>s\r\n" + A[3:70] + b"\r\n", False),
    "blank sequence lines": (b">r\n\n>s\n" + A + b"\n>t\n\n>u\n" + A[2:50] + b"\n", False),
    "same read many times": ((b">r\n" + A + b"\n") * 40, False),
    "FASTQ, last record cut short": (b"@r\n" + A + b"\n+\n" + b"I" * len(A) + b"\n@s\n" + A[4:64] + b"\n+\n", True),
    "FASTQ, CRLF": (b"@r\r\n" + A + b"\r\n+\r\n" + b"I" * len(A) + b"\r\n@s\r\n" + A[7:77] + b"\r\n+\r\n" + b"I" * 70 + b"\r\n", True),
    "lower case": (b">r\n" + A.lower() + b"\n>s\n" + A + b"\n>t\n" + A[:40] + A[40:].lower() + b"\n", False),
}
bad = 0
for name, (text, fastq) in CASES.items():
    for paired, clean in ((False, True), (True, True), (False, False), (True, False)):
        with tempfile.TemporaryDirectory() as td:
            p = os.path.join(td, "in.fq" if fastq else "in.fa")
            open(p, "wb").write(text)
            args = ["-size_kmer", "21", "-max_read_length", "100", "-estimated_kmers", "20000", "-singletons", "4000"] + (["--fastq"] if fastq else []) + (["--paired_ends"] if paired else []) + ([] if clean else ["--no_cleaning"])
            res = {}
            for tag, exe in (("ref", REF), ("gpu", EXE)):
                d = os.path.join(td, tag)
                os.mkdir(d)
                try:
                    res[tag] = subprocess.run(["stdbuf", "-o0", exe, "-read_load_file", p, "-read_scan_file", p, "-file_prefix", os.path.join(d, "out")] + args,
                                              capture_output=True, text=True, errors="replace", timeout=120)
                except subprocess.TimeoutExpired:
                    res[tag] = None
            files = sorted(os.listdir(os.path.join(td, "gpu")))
            rfiles = sorted(os.listdir(os.path.join(td, "ref")))
            ok = res["gpu"] is not None and res["gpu"].returncode in (0, 3)
            diff, cut = [], []
            for fn in files:
                a, b_ = os.path.join(td, "gpu", fn), os.path.join(td, "ref", fn)
                same = os.path.exists(b_) and open(a, "rb").read() == open(b_, "rb").read()
                if not same and os.path.exists(b_) and res["ref"] is not None and res["ref"].returncode < 0:
                    # the reference crashed in its contig-graph stage (no nodes to work on) with this file still in a stream's buffer: what it
                    # had written by then must be the head of ours
                    rb = open(b_, "rb").read()
                    same = open(a, "rb").read()[:len(rb)] == rb
                    cut.append((fn, len(rb)))
                if not same:
                    diff.append((fn, os.path.getsize(a), os.path.getsize(b_) if os.path.exists(b_) else None))
                ok = ok and same
            note = "" if ok else f" gpu rc {res['gpu'].returncode if res['gpu'] else 'timeout'} ref rc {res['ref'].returncode if res['ref'] else 'timeout'} differing {diff} ref stdout tail {res['ref'].stdout[-150:] if res['ref'] else ''!r}"
            print(f"{name}{', paired' if paired else ''}{'' if clean else ', --no_cleaning'}: {'equal' if ok else 'DIFFERENT'} ({len(files)} files{'; the reference crashed later, having written ' + str(cut) if cut else ''}){note}", flush=True)
            bad += 0 if ok else 1
print("failures:", bad)
sys.exit(1 if bad else 0)
