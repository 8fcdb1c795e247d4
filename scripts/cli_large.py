"""the `faucet` command line, file to files, at the size of BASELINE configs 5 and 4 on ONE GPU (GPU box; measurement).

The reads of tests/golden/fullsize.json's fixture (faucet_amd/synth_det.py) are written as a FASTA file into tmpfs chunk by chunk, `faucet` runs on
it with the phase clock on (FGPU_CLI_TIMES=1), and the line printed says per pass: wall time, GB/s of text consumed, the share of the pass the
device would have needed with the reads resident (the committed bench line's full_size.<config> step: what is left is the device waiting for text),
beside the page-locked host-to-device copy rate of this box.  Junction count and bloom weight are checked against the fixture's counters.
    python scripts/cli_large.py config5|config4 [resident_step_seconds]"""
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from faucet_amd import synth_det as sd  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "config5"
fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize.json")))[name]
c = fx["params"]
dev = torch.device("cuda", 0)
n, L_ = c["reads"], c["read_len"]
rec_bytes = 10 + L_ + 1
need = n * rec_bytes
base = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
free = shutil.disk_usage(base).free
if free < need + (8 << 30):
    print(json.dumps({"config": name, "skipped": f"{base} has {free >> 30} GiB free, the file needs {need >> 30} GiB"}))
    sys.exit(0)
d = os.path.join(base, "faucet_cli_large")
shutil.rmtree(d, ignore_errors=True)
os.makedirs(d)
path = os.path.join(d, name + ".fa")
t0 = time.perf_counter()
genome = sd.make_genome(c["genome"], c["genome_seed"], dev)
chunk = 4_000_000
with open(path, "wb") as f:
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        host = sd.make_reads(genome, m, L_, c["err"], c["read_seed"], dev, first_row=lo).cpu().numpy()
        rec = np.empty((m, rec_bytes), dtype=np.uint8)
        rec[:, 0] = ord(">")
        idx = np.arange(lo, lo + m, dtype=np.int64)
        for dgt in range(8):
            rec[:, 8 - dgt] = ord("0") + (idx // 10 ** dgt) % 10
        rec[:, 9] = ord("\n")
        rec[:, 10:10 + L_] = host
        rec[:, 10 + L_] = ord("\n")
        rec.tofile(f)
        print(f"wrote {lo + m} of {n} reads ({time.perf_counter() - t0:.0f} s)", file=sys.stderr, flush=True)
del genome
torch.cuda.empty_cache()
size = os.path.getsize(path)
# the page-locked copy rate of this box (what a pass that only moved the text would run at)
pin = torch.empty(1 << 30, dtype=torch.uint8, pin_memory=True)
dst = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
dst.copy_(pin, non_blocking=True)
torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(4):
    dst.copy_(pin, non_blocking=True)
torch.cuda.synchronize()
h2d = 4 * (1 << 30) / (time.perf_counter() - t1) / 1e9
del pin, dst
torch.cuda.empty_cache()

exe = os.path.join(ROOT, "faucet_amd", "faucet")
cmd = [exe, "-read_load_file", path, "-read_scan_file", path, "-size_kmer", str(c["k"]), "-max_read_length", str(L_), "-estimated_kmers", str(c["E"]),
       "-singletons", str(c["S"]), "--no_cleaning", "-file_prefix", os.path.join(d, "out")] + os.environ.get("CLI_EXTRA", "").split()
runs = []
for _ in range(2):
    t2 = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, FGPU_CLI_TIMES="1"))
    dt = time.perf_counter() - t2
    if r.returncode != 0:
        print(json.dumps({"config": name, "error": "faucet exited with %d: %s" % (r.returncode, r.stderr[-500:])}))
        sys.exit(1)
    ph = {m.group(1).strip(): float(m.group(2)) for m in re.finditer(r"\[cli\] (pass [12][^\d]*?)\s+([0-9.]+) ms", r.stderr)}
    waits = re.search(r"pass 2: the host waited for the device (\d+) times, ([0-9.]+) ms", r.stderr)
    runs.append({"seconds": dt, "pass_ms": ph, "pass2_host_waits": int(waits.group(1)) if waits else None,
                 "phase_clock": [ln.strip() for ln in r.stderr.splitlines() if ln.startswith("[cli]")]})
best = min(runs, key=lambda x: x["seconds"])
mj = re.search(r"Distinct junctions: (\d+)", r.stdout)
kmers = n * (L_ - c["k"] + 1)
p1 = next((v for k2, v in best["pass_ms"].items() if k2.startswith("pass 1")), None)
p2 = next((v for k2, v in best["pass_ms"].items() if k2.startswith("pass 2")), None)
resident = float(sys.argv[2]) if len(sys.argv) > 2 else None
bloom_sha = hashlib.sha256()
with open(os.path.join(d, "out.bloom"), "rb") as f:
    for blk in iter(lambda: f.read(1 << 24), b""):
        bloom_sha.update(blk)
out = {"config": name, "extra_arguments": os.environ.get("CLI_EXTRA", ""), "input_bytes": size, "kmers": kmers, "seconds": best["seconds"], "value": kmers / best["seconds"], "unit": "k-mers/s",
       "pass_ms": best["pass_ms"], "load_scan_value": kmers / ((p1 + p2) / 1e3) if p1 and p2 else None,
       "text_GBps": {"pass 1": size / (p1 / 1e3) / 1e9 if p1 else None, "pass 2": size / (p2 / 1e3) / 1e9 if p2 else None},
       "pinned_h2d_GBps_this_box": h2d, "pass2_host_waits": best["pass2_host_waits"], "both_runs_seconds": [x["seconds"] for x in runs],
       "junctions": int(mj.group(1)) if mj else None, "junctions_equal_the_oracles": bool(mj) and int(mj.group(1)) == int(fx["counters"]["n_junctions"]),
       "bloom_equals_the_oracles": bloom_sha.hexdigest() == fx["bloo2_sha256"],
       "resident_step_seconds": resident, "phase_clock": best["phase_clock"],
       "device_share_of_the_passes": resident / ((p1 + p2) / 1e3) if resident and p1 and p2 else None,
       "note": "wall time of the whole `faucet` process on a FASTA file in tmpfs (start-up, both passes, .bloom and .junctions written), best of two runs; "
               "device_share = the resident-input step of the same workload (bench line, full_size) / (pass 1 + pass 2): the rest is the device waiting for text"}
if os.environ.get("CLI_ROCPROF") == "1":      # one more run under rocprofv3 --stats: what the device spends per kernel with text as the input
    import csv
    import glob
    pd = os.path.join(d, "prof")
    subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", pd, "-o", "cli", "--"] + cmd, capture_output=True, text=True,
                   env=dict(os.environ, FGPU_CLI_TIDY="1", TMPDIR="/tmp"), cwd="/tmp")
    fs = glob.glob(os.path.join(pd, "**", "*kernel_stats.csv"), recursive=True)
    if fs:
        rows = list(csv.DictReader(open(fs[0])))
        out["kernels_ms"] = [[r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48], int(r["Calls"]), round(int(r["TotalDurationNs"]) / 1e6, 1)]
                             for r in rows[:22]]
        out["kernels_total_ms"] = round(sum(int(r["TotalDurationNs"]) for r in rows) / 1e6, 1)
print(json.dumps(out))
shutil.rmtree(d, ignore_errors=True)
