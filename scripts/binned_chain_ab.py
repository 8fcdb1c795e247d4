"""NS1 as a whole probe CHAIN (GPU box):   python scripts/binned_chain_ab.py
Bloom::contains (n_hash dependent bit tests with early exit) for 2^28 items, directly and with the first level binned by filter slice, the survivors
handed back as a dense list and the rest of their chains run directly (fgpu_diag_binned_chain) -- the shape a binned first level of
k_scan_flags_sm would have.  Filter sizes of the configurations, the fill of a filter at work (3 of 8 bits), 2 and 3 hash functions."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faucet_amd import api  # noqa: E402

ctx = api.Context(31, 1 << 29, 3)
n = 1 << 28
print(f"{n} chains per measurement (one MI355X)")
for table in (64 << 20, 512 << 20):
    for nh, fill in ((3, 0x29), (3, 0x5A), (2, 0x29)):
        for sl in (2 << 20, 4 << 20, 8 << 20):
            if table // sl > 512:
                continue
            r = ctx.diag_binned_chain(table, n, sl, nh, fill, 3)
            binned = r["bin_ms"] + r["first_ms"] + r["rest_ms"]
            print(f"table {table >> 20:4d} MiB, {nh} hashes, fill {bin(fill).count('1')}/8, slices of {sl >> 20} MiB: direct {r['direct_ms']:.2f} ms | binned "
                  f"{binned:.2f} ms = bin {r['bin_ms']:.2f} + first level {r['first_ms']:.2f} + survivors' chains {r['rest_ms']:.2f} "
                  f"({100 * r['survivors_share']:.0f} % survive) | direct/binned {r['direct_ms'] / binned:.2f}x | answers equal: {r['equal']}", flush=True)
