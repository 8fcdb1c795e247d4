"""Digest of the synth_det reads of a full-size case, on whatever device is there (CPU here, GPU on the box): the two must agree
before any full-size fixture means anything.   python scripts/det_digest.py config2 [config3 config5]"""
import hashlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import make_fullsize as mf  # noqa: E402

dev = "cuda" if torch.cuda.is_available() else "cpu"
for name in sys.argv[1:]:
    t0 = time.time()
    r = mf.make_case_reads(mf.CASES[name], dev)
    h = hashlib.sha256(r.cpu().numpy().tobytes()).hexdigest()
    print(name, dev, tuple(r.shape), h, f"{time.time() - t0:.1f} s", flush=True)
