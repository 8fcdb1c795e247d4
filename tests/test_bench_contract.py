"""The bench line committed under profiles/ carries every key of the driver's contract (CPU-only check of the evidence file)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_keys():
    d = json.loads(open(os.path.join(ROOT, "profiles", "r06_bench_config2.json")).read().strip().splitlines()[-1])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == base["metric"] and d["unit"] == "k-mers/s"
    for key in ("value", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["traffic"] is not None
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] == 1 and c["value"] > 0 and "sample" in c
    # since round 2: what the counters say, the PCIe-inclusive rate and the all-cores CPU figure ride in the same line
    assert d["pipeline_measured"]["GBps"] > 0 and not d["pipeline_measured"]["kernels_without_counters"]
    assert 0 < d["host_input"]["value"] < d["value"] and c["all_cores"]["cores"] > 1 and c["all_cores"]["value"] > c["value"]
    f = d["cli_file_to_files"]      # the command line on the same reads as a file: slower than the resident step, same junctions
    assert 0 < f["value"] < d["host_input"]["value"] and f["junctions_equal_the_steps"] and f["input_bytes"] > 10 ** 9
    s3 = d["stage3_find_neighbors"]   # findNeighbor from every junction of the step's own map, whole walks on the device
    assert s3["walks"] > d["outputs"]["junctions"] and s3["probes"] > s3["walks"] and s3["asserts_tripped"] == 0 and s3["value"] > 1e8
    # round 3: the headline fraction charges one sector per access the kernel really makes and has to agree with what the counters
    # measured for the same kernel; the reference-accesses figure (round 2's headline) only rides along
    assert "attribution" in r and r["frac"] < r["frac_reference_accesses"]
    # round 5: the same kernel on the large filters rides along, the timing method is stated, the C++ host over two read shards writes the same files
    rl = d["roofline_large"]
    assert rl["kernel"] == "load_mark" and 0 < rl["frac"] < r["frac"] and abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-9
    # round 6: ... with ITS counters (profiles/pmc_traffic.json, full_size.config4): traffic close to the algorithmic bytes, no wasted re-reads
    assert rl["traffic"] and 1.0 <= rl["traffic_over_algorithmic"] < 1.5 and abs(rl["frac_measured_traffic"] - rl["traffic"] / (rl["avg_launch_ms"] * 1e-3) / 1e9 / rl["peak"]) < 1e-9
    assert d["rccl_ranks"] == 0 and d["hbm_per_rank"][0]["used_bytes_after_the_steps"] > 0
    assert d["ms_per_step_with_events"] > 0 and "without" in d["timing_method"].lower()
    g2 = f["gpus2_one_device"]
    assert all(g2["files_equal_the_single_device_runs"].values()) and g2["seconds"] > 0
    assert abs(r["frac"] - r["frac_measured_traffic"]) / r["frac_measured_traffic"] < 0.25
    assert abs(r["frac_measured_traffic"] - r["traffic"] / (r["avg_launch_ms"] * 1e-3) / 1e9 / r["peak"]) < 1e-9
    c3 = d["config3_cli"]            # the slowest configuration, file to files, in the driver's line; every file equal to the compiled reference's
    assert c3["junctions_equal_the_references"] and all(c3["files_equal_the_references"].values()) and len(c3["files_equal_the_references"]) == 4
    assert 0 < c3["value"] < f["value"] and c3["kmers"] == 350_000_000
    # round 4: config 3 by SURVEY 8d's definition of the metric (pass 1 + pass 2 on the CLI's own clock) is above the 1e9 bar, and the line says so
    p = c3["pass_ms"]
    assert abs(c3["load_scan_value"] - c3["kmers"] / ((p["pass 1 (read + load)"] + p["pass 2 (read + scan)"]) / 1e3)) / c3["load_scan_value"] < 1e-9
    assert c3["load_scan_value"] > 1e9
    # round 4: the headline steps run without HIP events; kernel times and the roofline's launch time come from separate bracketed steps
    ps = d["profiled_steps"]
    # (the bracketed steps cost ~1.5 % more by an A/B on one box; run to run on the pool either group of steps may come out a few per cent ahead)
    assert ps["steps"] >= 1 and ps["ms_per_step"] >= 0.95 * d["ms_per_step"]
    assert abs(d["roofline"]["avg_launch_ms"] * d["roofline"]["launches"] / ps["steps"] - d["kernel_ms_per_step_rank0"]["load_mark"]) < 0.01
    for name in ("config5", "config4"):
        assert d["full_size"][name]["counters_equal_the_oracles"] is True and d["full_size"][name]["value"] > 4e9
    # value is consistent with the step time it was derived from
    assert abs(d["value"] - d["kmers_per_step"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6


def test_batch_bounds_cover_the_reads_once_in_order():
    """bench.py's batching (host logic): contiguous, complete, in file order; the ramp starts and ends small"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for n, batch, ramp in ((10_000_000, 1_000_000, 2), (10_000_000, 1_000_000, 0), (10_000_003, 1_000_000, 3), (3_000_000, 1_000_000, 2),
                           (999, 1000, 2), (25_000_000, 1_000_000, 5), (7, 2, 1)):
        b = bench.batch_bounds(n, batch, ramp)
        assert b[0][0] == 0 and b[-1][1] == n and all(x[1] == y[0] for x, y in zip(b[:-1], b[1:])) and all(lo < hi for lo, hi in b)
        assert max(hi - lo for lo, hi in b) <= batch
    sizes = [hi - lo for lo, hi in bench.batch_bounds(10_000_000, 1_000_000, 2)]
    assert sizes[:3] == [250_000, 250_000, 500_000] and sizes[-2:] == [500_000, 250_000] and sizes.count(1_000_000) == 8


def test_design_md_quotes_the_committed_bench_line():
    """DESIGN.md's number tables are printed from profiles/r06_bench_config2.json by scripts/design_tables.py (one number per
    configuration, one source): every generated table row and summary line has to be in the document as it is printed today."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "design_tables.py"), os.path.join("profiles", "r06_bench_config2.json")],
                         capture_output=True, text=True, cwd=ROOT, check=True).stdout
    doc = open(os.path.join(ROOT, "DESIGN.md")).read()
    rows = [ln for ln in out.splitlines() if ln.strip() and not set(ln) <= set("|- ")]
    assert len(rows) >= 12
    missing = [ln for ln in rows if ln not in doc and ("* " + ln) not in doc]
    assert not missing, missing
