"""Random-access ceilings of the device by table size (diagnostic; run on the GPU box)."""
from faucet_amd import api
ctx = api.Context(31, 1 << 29, 3)
print("stream copy GB/s", round(ctx.diag_stream_copy(1 << 30, 5)))
for lg in (20, 21, 22, 23, 24, 26, 27, 28, 29, 30, 31):
    r = [ctx.diag_random_access(1 << lg, 1 << 28, m, 3) for m in (0, 1, 2)]
    print(f"table 2^{lg} B: load {r[0]:.3e}/s  atomicMin {r[1]:.3e}/s  test+or {r[2]:.3e}/s")
