"""Prints the number tables of DESIGN.md (top of the document, section 7) from a committed bench line, so that the document quotes ONE number
per configuration and every number has a file behind it.
    python scripts/design_tables.py profiles/r04_bench_config2.json"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
src = sys.argv[1]
k = d["kernel_ms_per_step_rank0"]
r = d["roofline"]
c3 = d.get("config3_cli", {})
fs = d.get("full_size", {})
cli = d.get("cli_file_to_files", {})
cpu = d.get("cpu_baseline", {})
print(f"| configuration | seconds per step | k-mers/s | where in `{src}` |")
print("|---|---|---|---|")
print(f"| 2 (10 M × 100 bp, E = 1e8; the headline) | {d['ms_per_step'] / 1e3:.4f} | **{d['value']:.3g}** | `value`, `ms_per_step` ({d['steps']} steps, no events around the kernels) |")
if d.get("host_input"):
    print(f"| 2, reads handed over as host buffers (PCIe inside the step; never `value`) | {d['host_input']['ms_per_step'] / 1e3:.4f} | {d['host_input']['value']:.3g} | `host_input` |")
if cli:
    print(f"| 2 as a 1.1 GB FASTA file through the `faucet` process, start-up and output files included | {cli['seconds']:.3f} | {cli['value']:.3g} | `cli_file_to_files` |")
    g2 = cli.get("gpus2_one_device")
    if g2 and "seconds" in g2:
        print(f"| the same file through `faucet -gpus 2` (the C++ host over two read shards, BOTH on this box's one device: a functional record, not a scaling number; files equal: {all(g2['files_equal_the_single_device_runs'].values())}) | {g2['seconds']:.3f} | {g2['value']:.3g} | `cli_file_to_files.gpus2_one_device` |")
if c3:
    p = c3.get("pass_ms") or {}
    p1 = [v for n, v in p.items() if n.startswith("pass 1")]
    p2 = [v for n, v in p.items() if n.startswith("pass 2")]
    if p1 and p2:
        print(f"| 3's shape (2.5 M pairs × 100 bp, repeats, `--fastq --paired_ends`, cleaning), load + scan by SURVEY §8d's definition | "
              f"{(p1[0] + p2[0]) / 1e3:.3f} (pass 1 {p1[0]:.0f} ms + pass 2 {p2[0]:.0f} ms) | **{c3['load_scan_value']:.3g}** | `config3_cli.load_scan_value`, `.pass_ms` |")
    print(f"| 3's shape, the whole process (HIP start-up, four output files, exit) | {c3['seconds']:.3f} | {c3['value']:.3g} | `config3_cli.value` |")
for name, label in (("config5", "5 (50 M × 150 bp, 5 % errors, two hash functions, 2 × 1 GiB)"), ("config4", "4 whole on ONE GPU (200 M × 100 bp, 2 × 1 GiB filters)")):
    f = fs.get(name)
    if f and "value" in f:
        print(f"| {label} | {f['seconds']:.3f} | **{f['value']:.3g}** | `full_size.{name}` (second step of the context; counters equal the oracle's: {f['counters_equal_the_oracles']}) |")
import os
for name, label in (("config5", "5 as an 8.05 GB FASTA file through the `faucet` process"), ("config4", "4 as a 22.2 GB FASTA file through the `faucet` process")):
    path = os.path.join(os.path.dirname(os.path.abspath(src)), f"r06_cli_large_{name}.json")
    if os.path.exists(path):
        c = json.load(open(path))
        print(f"| {label} (start-up, both passes, `.bloom` and `.junctions` of {c['junctions']:.3g} records written) | {c['seconds']:.2f} | {c['value']:.3g} | `profiles/r06_cli_large_{name}.json` (`scripts/cli_large.py`; passes alone: {c['load_scan_value']:.3g}) |")
print()
print("per kernel per step (ms, `kernel_ms_per_step_rank0`, from the bracketed steps): " + ", ".join(f"`{n}` {v:.1f}" for n, v in list(k.items())[:11]))
print(f"roofline ({r['kernel']}): {r['avg_launch_ms']:.3f} ms per launch of {r['kmers_per_launch']:.3g} k-mers, frac {r['frac']:.3f} (sectors needed), "
      f"{r['frac_measured_traffic']:.3f} (counters), {r['frac_reference_accesses']:.3f} (reference's separate arrays)")
rl = d.get("roofline_large")
if rl:
    print(f"roofline_large ({rl['kernel']} on config 4's 2 x 1 GiB filters): {rl['avg_launch_ms']:.2f} ms per launch of {rl['kmers_per_launch']:.3g} k-mers, frac {rl['frac']:.3f} (sectors needed)")
pa = d.get("pipeline_ab64", {})
pm = d.get("pipeline_measured", {})
if pa:
    print(f"pipeline AB64 {pa['bytes_per_kmer']:.0f} B per k-mer -> {pa['achieved_GBps'] / 1e3:.2f} TB/s = {pa['frac_of_hbm_peak']:.3f} of peak; "
          f"{pa.get('bit_accesses_per_s', 0):.3g} counted bit accesses/s = {pa.get('frac_of_random_access_ceiling', 0):.3f} of the measured ceiling; "
          f"by counters {pm.get('hbm_bytes_per_step', 0) / 1e9:.0f} GB per step = {pm.get('GBps', 0) / 1e3:.2f} TB/s = {pm.get('frac_of_hbm_peak', 0):.3f}")
if cpu:
    print(f"cpu: {cpu['kind']} {cpu['value']:.3g} k-mers/s on {cpu['cores']} core; port {cpu.get('port', {}).get('value', 0):.3g}; "
          f"all cores {cpu.get('all_cores', {}).get('value', 0):.3g} on {cpu.get('all_cores', {}).get('cores')}")
