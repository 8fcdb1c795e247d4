"""An extreme shape through the command line (GPU box): 180 000 read pairs of a 60 kb genome with twelve planted repeats -- 600x coverage, a 36 MB
file, three batches -- where the large clusters of a window outgrow the tables of the large-cluster walks as soon as the windows grow: pass 2's
time by window bound, and the window-size decisions with FGPU_DEBUG_SPAN.    python scripts/extreme_coverage_cli.py"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from faucet_amd import synth  # noqa: E402

g = synth.make_genome(60_000, 900, repeats=12, repeat_len=500)
r = synth.make_pairs(g, 180_000, 100, 300, 25, 0.01, 950)
with tempfile.TemporaryDirectory() as td:
    p = os.path.join(td, "in.fq")
    synth.write_fastq(p, r)
    args = ["-size_kmer", "21", "-max_read_length", "100", "-estimated_kmers", "400000", "-singletons", "80000", "--fastq", "--paired_ends"]
    for env in ({}, {}, {"FGPU_MAX_SPAN_LOG2": "20"}, {"FGPU_DEBUG_SPAN": "1"}):
        rr = subprocess.run([os.path.join(ROOT, "faucet_amd", "faucet"), "-read_load_file", p, "-read_scan_file", p, "-file_prefix", os.path.join(td, "o")] + args,
                            capture_output=True, text=True, env=dict(os.environ, FGPU_CLI_TIMES="1", **env))
        lines = [ln.strip() for ln in rr.stderr.splitlines() if "pass 2 (" in ln or "optimistically" in ln or "[span]" in ln]
        print(env or "defaults", "\n   " + "\n   ".join(lines), flush=True)
