"""Fills DESIGN.md's generated parts from the committed bench line and the strong-scaling projection: the text between `<!-- NAME -->` and `<!-- /NAME -->`
markers is replaced (TABLE1: one number per configuration; TABLE2: roofline / pipeline / cpu rows; SUMMARY: per-kernel lines of section 7; PROJTABLE: section 5).
    python scripts/design_fill.py profiles/r06_bench_config2.json profiles/r06_project_strong.txt"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
bench_path, proj_path = sys.argv[1], sys.argv[2]
out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "design_tables.py"), bench_path], capture_output=True, text=True, cwd=ROOT, check=True).stdout
table1, rest = out.split("\n\n", 1)
lines = [ln for ln in rest.splitlines() if ln.strip()]
d = json.loads(open(os.path.join(ROOT, bench_path)).read().strip().splitlines()[-1])
fs4 = d.get("full_size", {}).get("config4", {})
r, rl = d["roofline"], d.get("roofline_large") or {}
pa, pm, cpu = d.get("pipeline_ab64", {}), d.get("pipeline_measured", {}), d.get("cpu_baseline", {})
table2 = "\n".join([
    f"| roofline, dominant kernel `k_load_mark` | {r['avg_launch_ms']:.3f} ms per launch of {r['kmers_per_launch']:.3g} k-mers: **{r['frac']:.3f}** of the 8 TB/s peak on the sectors it needs (192 B per k-mer), "
    f"{r['frac_measured_traffic']:.3f} by the counters (FETCH+WRITE, `profiles/pmc_traffic.json`), {r['frac_reference_accesses']:.3f} on the reference's separate-array accesses; on config 4's 2 x 1 GiB filters "
    f"(`roofline_large`) {rl.get('avg_launch_ms', 0):.2f} ms per {rl.get('kmers_per_launch', 0):.3g} k-mers: **{rl.get('frac', 0):.3f}**, {rl.get('frac_measured_traffic') or 0:.3f} by ITS counters "
    f"({rl.get('traffic_over_algorithmic') or 0:.2f} x the algorithmic bytes) |",
    f"| pipeline | AB64 {pa.get('bytes_per_kmer', 0):.0f} B per k-mer -> {pa.get('achieved_GBps', 0) / 1e3:.2f} TB/s = {100 * pa.get('frac_of_hbm_peak', 0):.1f} % of peak; {pa.get('bit_accesses_per_s', 0):.3g} of the reference's counted bit "
    f"accesses per second = {100 * pa.get('frac_of_random_access_ceiling', 0):.0f} % of the device's measured random-access ceiling; by counters {pm.get('hbm_bytes_per_step', 0) / 1e9:.0f} GB per step = "
    f"{pm.get('GBps', 0) / 1e3:.2f} TB/s = {100 * pm.get('frac_of_hbm_peak', 0):.0f} % |",
    f"| CPU beside it | compiled reference {cpu.get('value', 0):.2g} k-mers/s on {cpu.get('cores', 1)} core, oracle port {cpu.get('port', {}).get('value', 0):.2g}, {cpu.get('all_cores', {}).get('cores')} replicas "
    f"{cpu.get('all_cores', {}).get('value', 0):.2g} (host cores differ from box to box) |"])
one = fs4.get("seconds")
rows, x8 = [], None
for ln in open(os.path.join(ROOT, proj_path)):
    m = re.match(r"N=(\d+): per rank (\d+) reads \| pass 1 \(([^)]*)\): (.*?) = (\d+) ms \| pass 2: rank 0 scan (\d+) ms, others' pure stage ([\d-]+) ms, hops \([^)]*\) (.*?) = (\d+) ms \| step (\d+) ms = ([\d.e+]+) k-mers/s", ln)
    if not m:
        continue
    n, per, proto, p1detail, p1, scan0, pure, hops, hopsum, step, rate = m.groups()
    speed = one * 1e3 / float(step) if one else 0.0
    if n == "8":
        x8 = speed
    hop = hops.split()[-1]
    rows.append((int(n), f"| {n} | {int(per) // 1000000} M | {proto}: {p1detail} = {p1} ms | {scan0} ‖ {pure} ms | ({hop.replace('+', ' + ')}) × {int(n) - 1} = {hopsum} ms | {float(step) / 1e3:.2f} s | {float(rate):.2g} | {speed:.2f}× |"))
proj = "\n".join(["| N | reads per rank | pass 1 (slowest rank) + exchanges | rank 0's scan ‖ others' pure stage | last hop: send + import/walk + export, × hops | step | k-mers/s | vs one GPU (" + (f"{one:.2f} s" if one else "?") + ") |",
                  "|---|---|---|---|---|---|---|---|"] + [r_ for _, r_ in sorted(rows)])
doc_path = os.path.join(ROOT, "DESIGN.md")
doc = open(doc_path).read()
for name, text in (("TABLE1", table1), ("TABLE2", table2), ("SUMMARY", "\n".join("* " + ln for ln in lines)), ("PROJTABLE", proj), ("PROJ8", f"{x8:.2f}" if x8 else "?")):
    inline = name == "PROJ8"
    pat = re.compile(r"<!-- %s -->.*?<!-- /%s -->" % (name, name), re.S)
    rep = f"<!-- {name} -->{text}<!-- /{name} -->" if inline else f"<!-- {name} -->\n{text}\n<!-- /{name} -->"
    if pat.search(doc):
        doc = pat.sub(lambda _m: rep, doc)
    else:
        doc = doc.replace("@@%s@@" % name, rep)
open(doc_path, "w").write(doc)
print("DESIGN.md:", len(doc.encode()), "bytes")
