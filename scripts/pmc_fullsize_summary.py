"""FETCH_SIZE / WRITE_SIZE per launch of every kernel of a full-size configuration (scripts/profile_fullsize.sh <config> <tag> pmc), written to
gpurun_out/prof_<config>/pmc_<config>.json; `python scripts/pmc_fullsize_summary.py --merge gpurun_out/prof_config4/pmc_config4.json ...` puts
such files into profiles/pmc_traffic.json under full_size.<config> (what bench.py's roofline_large.traffic reads)."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()


def counters(d, counter):
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            n = row["Kernel_Name"]
            if row["Counter_Name"] != counter or "k_" not in n or "at::" in n or "rocprim" in n:
                continue
            a = acc.setdefault(short(n), [0, 0.0])
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    return acc


if sys.argv[1] == "--merge":
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    d = json.load(open(path))
    for f in sys.argv[2:]:
        e = json.load(open(f))
        d.setdefault("full_size", {})[e["config"]] = e
    json.dump(d, open(path, "w"), indent=1)
    print("merged", list(d["full_size"]))
    sys.exit(0)

src, cfg = sys.argv[1], sys.argv[2]
fetch, write = counters(os.path.join(src, "pmc_FETCH_SIZE"), "FETCH_SIZE"), counters(os.path.join(src, "pmc_WRITE_SIZE"), "WRITE_SIZE")
kernels = {}
for k in sorted(set(fetch) | set(write)):
    fl, fs = fetch.get(k, [0, 0.0])
    wl, ws = write.get(k, [0, 0.0])
    kernels[k] = {"launches": max(fl, wl), "fetch_KB_per_launch": fs / fl if fl else None, "write_KB_per_launch": ws / wl if wl else None,
                  "bytes_per_launch": 1024.0 * ((fs / fl if fl else 0) + (ws / wl if wl else 0))}
step = None
try:
    step = json.loads(open(os.path.join(src, "step_FETCH_SIZE.json")).read().strip().splitlines()[-1])
except Exception:  # noqa: BLE001
    pass
out = {"config": cfg, "how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes over `python3 scripts/fullsize_step.py " + cfg +
                             "` (two steps of the full-size workload; scripts/profile_fullsize.sh " + cfg + " r06 pmc); bytes = (FETCH + WRITE) x 1024 per launch, "
                             "no x2 correction (4/8-byte random accesses, see the calibration entry of the config 2 passes)",
       "kernels": kernels, "kmers": step.get("kmers") if step else None}
json.dump(out, open(os.path.join(src, f"pmc_{cfg}.json"), "w"), indent=1)
for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["bytes_per_launch"] * kv[1]["launches"])[:12]:
    print(f"{k[:40]:40s} x{v['launches']:>6d} {v['bytes_per_launch'] / 1e9:9.3f} GB per launch")
