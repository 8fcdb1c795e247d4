"""Development aid: turn the CHECKPOINT lines of a tests/golden/make_fullsize.py config4 run that is still under way into
tests/golden/fullsize_partial.json, which tests/test_gpu_fullsize.py reads when fullsize.json has no config4 entry yet (the finished run
writes the real entry; the partial file is not committed)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402

log = sys.argv[1] if len(sys.argv) > 1 else "/tmp/config4_oracle.log"
params = dict(genome=400_000_000, genome_seed=4, reads=200_000_000, read_len=100, err=0.01, read_seed=4000, k=31,
              E=1_000_000_000, S=200_000_000, slice=5_000_000, shards=8)
tai, nh, _, _ = po.sizing_from_cli(params["E"], params["S"])
entry = {"params": params, "tai": tai, "n_hash": nh, "load_checkpoints": [], "scan_checkpoints": []}
for line in open(log):
    if line.startswith("CHECKPOINT "):
        _, which, js = line.split(" ", 2)
        entry[which + "_checkpoints"].append(json.loads(js))
out = os.path.join(ROOT, "tests", "golden", "fullsize_partial.json")
json.dump(entry, open(out, "w"), indent=1)
print(out, len(entry["load_checkpoints"]), "load and", len(entry["scan_checkpoints"]), "scan checkpoints")
