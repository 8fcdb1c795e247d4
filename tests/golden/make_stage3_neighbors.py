#!/usr/bin/env python3
"""Known answers of the reference's OWN JunctionMap::findNeighbor (utils/JunctionMap.cpp:231-412) on the golden junction maps:
    make -C oracle ref && python tests/golden/make_stage3_neighbors.py
For every junction of a case's `.junctions` (reloaded with JunctionMap::buildFromFile) and every extension a contig would be built on,
oracle/_ref/ref_kat neighbors runs findNeighbor against the case's `.bloom` and prints the result; stored as stage3_neighbors_<case>.jsonl.gz."""
import gzip
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests.golden_util import Case  # noqa: E402

KAT = os.path.join(ROOT, "oracle", "_ref", "ref_kat")
for name in ("c1_k21", "ragged_k31", "twohash_k31_L150", "j2_spacer20_k15", "j0_k15"):
    c = Case(name)
    with tempfile.TemporaryDirectory() as td:
        b, j = os.path.join(td, "b.bloom"), os.path.join(td, "j.junctions")
        open(b, "wb").write(c.bloom().tobytes())
        open(j, "w").write("\n".join(c.junction_lines()) + "\n")
        tai = len(c.bloom()) * 8
        r = subprocess.run([KAT, "neighbors", b, str(tai - 1), str(c.counters["n_hash"]), str(c.k), str(c.j), j, str(c.max_read_length)],
                           capture_output=True, text=True)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"kat":"neighbor"')]
        assert r.returncode == 0 and lines, (name, r.returncode, r.stderr[-500:])
        with gzip.GzipFile(os.path.join(HERE, f"stage3_neighbors_{name}.jsonl.gz"), "wb", mtime=0) as f:
            f.write(("\n".join(lines) + "\n").encode())
        print(name, len(lines), "calls,", sum('"abort"' in ln for ln in lines), "aborted by the reference's asserts")
