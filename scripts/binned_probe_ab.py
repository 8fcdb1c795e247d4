"""NS1's query-side blocking as an A/B (GPU box):   python scripts/binned_probe_ab.py
direct random probes vs probes binned by filter slice (fgpu_diag_binned_probes), for the filter sizes of the configurations and several slice sizes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faucet_amd import api  # noqa: E402

ctx = api.Context(31, 1 << 29, 3)
n = 1 << 28
print(f"{n} single-bit probes per measurement (one MI355X); a direct probe moves one 64-byte sector")
for table in (8 << 20, 64 << 20, 512 << 20):
    for sl in (1 << 20, 2 << 20, 4 << 20, 8 << 20):
        if sl > table or table // sl > 512:
            continue
        r = ctx.diag_binned_probes(table, n, sl, 3)
        print(f"table {table >> 20:4d} MiB, slices of {sl >> 20} MiB ({table // sl:3d}): direct {r['direct_per_s']:.3g}/s ({1e3 * n / r['direct_per_s']:.2f} ms) | binned {r['binned_per_s']:.3g}/s = "
              f"bin {r['bin_ms']:.2f} ms + probe {r['probe_ms']:.2f} ms | binned/direct {r['binned_per_s'] / r['direct_per_s']:.2f}x", flush=True)
