"""Random-access ceilings of the device by table size: the rate the filter kernels are held against (bench.py's `ceilings` measures them at
config 2's sizes) for the larger filters of BASELINE configs 4 and 5 (2 x 1 GiB interleaved = 2 GiB, 32 GiB of first-set times).
    gpurun -- python scripts/random_access_by_table_size.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faucet_amd import api  # noqa: E402

ctx = api.Context(31, 1 << 29, 3)
print("table bytes        load32/s   test+atomicOr32/s   atomicMin32/s")
for lg in (26, 27, 28, 29, 30, 31, 33, 35):
    tb = 1 << lg
    row = []
    for mode in (0, 2, 1):
        try:
            row.append(f"{ctx.diag_random_access(tb, 1 << 28, mode, 2):.3g}")
        except Exception as e:   # noqa: BLE001
            row.append("failed " + repr(e)[:40])
    print(f"2^{lg} = {tb >> 20:6d} MiB   " + "   ".join(f"{r:>12s}" for r in row), flush=True)
