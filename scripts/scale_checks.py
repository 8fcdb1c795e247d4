"""Larger-than-suite invariance checks (diagnostic; GPU box, ~1 minute):
  * scanInputRead's lists (fgpu_scan_take_stops) for 10 M reads are the same with lazy and eager junction tests and under another batching;
  * --mercy on 1 M low-coverage reads equals the oracle's mercy load."""
import hashlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from faucet_amd import _lib as L  # noqa: E402
from faucet_amd import api  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

dev = torch.device("cuda", 0)


def stops_digest(reads, tai, nh, b2, batch, eager):
    ctx = api.Context(31, tai, nh, record_stops=True, eager_flags=eager)
    ctx.bloom_upload(L.BLOO2, b2)
    batches = bench.device_batches(reads, batch)
    h = hashlib.sha256()
    total, base = 0, []
    lo = 0
    for b in batches:
        base.append(lo)
        lo += b.n_reads
    ctx.scan_begin()
    for b in batches:
        ctx.scan_batch(b)
    st = ctx.scan_end()
    while True:
        t = ctx.take_stops()
        if t is None:
            break
        seq, s = t
        rec = np.empty(len(s), dtype=[("read", "<u8"), ("ext", "<u8"), ("info", "<u4")])   # one record per element: batching-free
        rec["read"] = s["read"].astype(np.uint64) + np.uint64(base[seq])                  # global read index
        rec["ext"], rec["info"] = s["ext"], s["info"]
        h.update(rec.tobytes())
        total += len(s)
    ctx.close()
    return h.hexdigest()[:16], total, st["n_junctions"]


n = 10_000_000
reads = bench.make_reads(bench.make_genome(2 * n, 2, dev), n, 100, 0.01, 1000, dev)
tai, nh = api.load_filter_shape(10 * n, 2 * n)
ctx = api.Context(31, tai, nh)
lst, sst, b2, keys, recs = bench.step_single(ctx, bench.device_batches(reads, 1_000_000))
ctx.close()
a = stops_digest(reads, tai, nh, b2, 1_000_000, False)
b = stops_digest(reads, tai, nh, b2, 1_000_000, True)
c = stops_digest(reads, tai, nh, b2, 700_000, False)
print("stops lazy", a, "eager", b, "other batching", c, "->", "PASS" if a == b == c else "FAIL", flush=True)

from faucet_amd import synth  # noqa: E402
g = synth.make_genome(8_000_000, 77)
r = synth.make_reads(g, 1_000_000, 100, 0.01, 78, n_rate=0.0005)
bases, offs = po.reads_from_matrix(r)
tai2, nh2 = api.load_filter_shape(20_000_000, 8_000_000)
ob1, ob2 = po.Bloom(tai2, nh2), po.Bloom(tai2, nh2)
po.load_two_filters(ob1, ob2, bases, offs, 31, mercy=True)
ctx = api.Context(31, tai2, nh2, mercy=True)
parts = [api.ReadBatch(bases, offs[a:min(a + 130_000, 1_000_000) + 1].copy()) for a in range(0, 1_000_000, 130_000)]
ctx.load_begin()
for p in parts:
    ctx.load_batch(p)
ctx.load_end()
ok = np.array_equal(ctx.bloom_download(L.BLOO2), ob2.bits()) and np.array_equal(ctx.bloom_download(L.BLOO1), ob1.bits())
plain1, plain2 = po.Bloom(tai2, nh2), po.Bloom(tai2, nh2)
po.load_two_filters(plain1, plain2, bases, offs, 31)
extra = int(np.unpackbits(ob2.bits()).sum() - np.unpackbits(plain2.bits()).sum())
print("mercy 1 M reads vs oracle:", "PASS" if ok else "FAIL", "(mercy adds", extra, "bits over the plain load)", flush=True)
