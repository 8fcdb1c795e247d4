"""Turn the rocprofv3 outputs of scripts/profile_round.sh into the summaries kept under profiles/.

    python scripts/pmc_summary.py gpurun_out/prof_r01 r01

writes profiles/<tag>_rocprofv3_kernel_stats.csv (our kernels only) and profiles/pmc_traffic.json
(FETCH_SIZE / WRITE_SIZE per launch for every fgpu kernel, plus the calibration of the counters on the
random-access diagnostic kernel whose byte count is known).
"""
import csv
import glob
import json
import os
import sys


def our(name):
    return "k_" in name and "at::" not in name and "rocprim" not in name


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0].strip()


def counters(d, counter):
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter or not our(row["Kernel_Name"]):
                continue
            a = acc.setdefault(short(row["Kernel_Name"]), [0, 0.0])
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    return acc


def main():
    src, tag = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # ---- kernel statistics
    rows = []
    for f in glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True):
        rd = csv.DictReader(open(f))
        fields = rd.fieldnames
        for row in rd:
            if our(row["Name"]):
                row["Name"] = short(row["Name"])
                rows.append(row)
    if rows:
        with open(os.path.join(root, "profiles", f"{tag}_rocprofv3_kernel_stats.csv"), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=fields)
            w.writeheader()
            w.writerows(rows)
    # ---- PMC traffic
    fetch, write = counters(os.path.join(src, "pmc_fetch"), "FETCH_SIZE"), counters(os.path.join(src, "pmc_write"), "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        fl, fs = fetch.get(k, [0, 0.0])
        wl, ws = write.get(k, [0, 0.0])
        kernels[k] = {"launches": max(fl, wl), "fetch_KB_per_launch": fs / fl if fl else None, "write_KB_per_launch": ws / wl if wl else None}
    cfg = json.load(open(os.path.join(src, "bench_fetch.json")))
    out = {
        "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 bench.py --steps 1 --warmup 0 --no-cpu` "
               "(scripts/profile_round.sh); bytes = (FETCH_SIZE + WRITE_SIZE) * 1024 averaged per launch.  No x2 correction is applied to FETCH_SIZE: "
               "that correction (MI355X_MICROARCH.md, HBM) is for wide coalesced streams, these kernels issue 4/8-byte random accesses; the "
               "calibration entry shows what the counter reports per access for bare random 4-byte loads of a known count.",
        "workload": cfg["config"]["workload"], "batch_reads": cfg["config"]["batch_reads"],
        "kernels": kernels,
        "bytes_per_launch": {k: 1024.0 * ((v["fetch_KB_per_launch"] or 0) + (v["write_KB_per_launch"] or 0)) for k, v in kernels.items()},
    }
    n_acc = cfg.get("ceilings", {}).get("accesses_per_measurement")
    cal = {}
    for k, v in kernels.items():
        if "k_diag_random" in k and n_acc:
            cal[k] = {"fetch_bytes_per_access": 1024.0 * (v["fetch_KB_per_launch"] or 0) / n_acc, "write_bytes_per_access": 1024.0 * (v["write_KB_per_launch"] or 0) / n_acc}
    out["calibration"] = cal
    try:      # (the counters of configs 4 and 5 at full size are another script's: scripts/pmc_fullsize_summary.py)
        old = json.load(open(os.path.join(root, "profiles", "pmc_traffic.json")))
        if "full_size" in old:
            out["full_size"] = old["full_size"]
    except (OSError, ValueError):
        pass
    with open(os.path.join(root, "profiles", "pmc_traffic.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: v for k, v in out.items() if k in ("calibration",)}, indent=1))
    for k, v in out["bytes_per_launch"].items():
        print(f"{k:40s} {v / 1e6:12.1f} MB/launch  x{kernels[k]['launches']}")


if __name__ == "__main__":
    main()
