"""bench.py's full-size leg of one BASELINE configuration on its own (two steps of the committed fixture's reads, counters checked against the
oracle's), for profilers:   rocprofv3 --kernel-trace --stats ... -- python3 scripts/fullsize_step.py config5"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "config5"
br = {"config5": 2_000_000, "config4": 2_500_000}.get(name, 2_000_000)
print(json.dumps(bench.full_size_leg(name, torch.device("cuda", 0), br)))
