"""Does a second process's idle GPU context slow the command line down?  (GPU box.)  config 3's shape through the command line, first while THIS
process holds a library context of config 2's size and has run a step on it (what bench.py's CLI legs used to run beside), then with that
context closed and torch's cache emptied.  Pass times by the CLI's own clock."""
import json
import os
import re
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from faucet_amd import api  # noqa: E402
from faucet_amd import synth_det as sd  # noqa: E402

fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize.json")))["config3"]
c = fx["params"]
dev = torch.device("cuda", 0)
g = sd.make_genome(c["genome"], c["genome_seed"], dev)
sd.plant_repeats(g, c["genome_seed"] + 100, *c["repeats"])
reads3 = sd.make_pairs(g, c["pairs"], c["read_len"], c["insert"][0], c["insert"][1], c["err"], c["read_seed"], dev)
sd.fasta_bytes(reads3, fastq=True).cpu().numpy().tofile("/dev/shm/c3b_reads.fq")
del reads3, g


def cli(tag):
    r = subprocess.run([os.path.join(ROOT, "faucet_amd", "faucet"), "-read_load_file", "/dev/shm/c3b_reads.fq", "-read_scan_file", "/dev/shm/c3b_reads.fq",
                        "-file_prefix", "/dev/shm/c3b_out"] + fx["args"], capture_output=True, text=True, env=dict(os.environ, FGPU_CLI_TIMES="1"))
    p = {m.group(1): float(m.group(2)) for m in re.finditer(r"\[cli\] (pass [12]) [^\d]*?([0-9.]+) ms", r.stderr)}
    print(f"{tag}: pass 1 {p.get('pass 1')} ms, pass 2 {p.get('pass 2')} ms", flush=True)


tai, nh = api.load_filter_shape(100_000_000, 20_000_000)
reads = bench.make_reads(bench.make_genome(20_000_000, 2, dev), 10_000_000, 100, 0.01, 1000, dev)
batches = bench.device_batches(reads, bench.batch_bounds(10_000_000, 1_000_000, 2))
ctx = api.Context(31, tai, nh)
bench.step_single(ctx, batches, pinned=True)
ctx.synchronize()
for i in range(4):
    cli("beside an idle context of this process")
ctx.close()
del reads, batches
torch.cuda.empty_cache()
for i in range(4):
    cli("with that context closed")
for f in os.listdir("/dev/shm"):
    if f.startswith("c3b_"):
        os.remove(os.path.join("/dev/shm", f))
