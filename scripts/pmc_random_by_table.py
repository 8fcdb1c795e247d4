"""The program scripts/pmc_random_by_table.sh profiles: k_diag_random (2^28 independent 4-byte accesses) as loads and as atomicMin on tables of
(256 MiB with FGPU_PMC_WITH_256MIB=1: what the Infinity Cache holds,) 2 GiB and 32 GiB -- config 2's and config 4's first-set-time tables -- in a fixed order, one launch each, so the counter rows can be told apart
by dispatch order.  Prints the rate of every case."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faucet_amd import api  # noqa: E402

ctx = api.Context(31, 1 << 29, 3)
for lg in ((28, 31, 35) if os.environ.get('FGPU_PMC_WITH_256MIB') else (31, 35)):
    for mode in (0, 1):
        rate = ctx.diag_random_access(1 << lg, 1 << 28, mode, 1)
        print(f"table 2^{lg} B mode {('load32', 'atomicMin32')[mode]} {rate:.4g} /s", flush=True)
