import sys
sys.path.insert(0, "/root/repo")
from faucet_amd import api
ctx = api.Context(31, 1 << 29, 3)
for lg in (28, 31, 33, 34, 35):
    tb = 1 << lg
    for mode, name in ((0, "load"), (1, "atomicMin")):
        try:
            r = ctx.diag_random_access(tb, 1 << 28, mode, 2)
            print(f"table 2^{lg} B ({tb >> 30} GiB) {name:10s} {r:.3g}/s", flush=True)
        except Exception as e:
            print(lg, name, "failed", e)
