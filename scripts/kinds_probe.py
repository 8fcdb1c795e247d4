"""`k_load_mark` on 2^33-bit filters comes in "two kinds" from run to run.  What decides the kind -- the process, the context, the
allocation?  Pass 1 alone on BASELINE config 4's per-GPU shape (25 M reads, 2 x 1 GiB filters; scripts/load_layout_ab.py's shape), load_mark per pass:
  mode procs      N processes, each: one context, 3 passes
  mode contexts   one process: 4 contexts one after the other (destroyed in between), 3 passes each -- does the kind change inside a process?
  mode order      one process, FGPU_DEBUG_ALLOC_FIRST=1 in the child: the first-set times (32 GiB) allocated before anything else of the context
GPU box.  usage: python scripts/kinds_probe.py procs|contexts|order [n]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one_context(reads, batches, tag):
    import torch
    from faucet_amd import api
    tai, nh = api.load_filter_shape(1_000_000_000, 200_000_000)
    free0, total = torch.cuda.mem_get_info(0)
    ctx = api.Context(31, tai, nh, profile=True)
    out = []
    for rep in range(3):
        ctx.kernel_times_reset()
        ctx.load_begin()
        for b in batches:
            ctx.load_batch(b)
        ctx.load_end()
        t = ctx.kernel_times()
        out.append(t.get("load_mark", (0, 0.0))[1])
    import ctypes as C
    pl, fa, mx = C.c_double(0), C.c_double(0), C.c_double(0)
    rc = ctx.lib.fgpu_diag_load_tables(ctx.h, 1 << 28, C.byref(pl), C.byref(fa), C.byref(mx))
    rates = f"pair loads {pl.value:.3e}/s, first[] atomicMin {fa.value:.3e}/s, mixed {mx.value:.3e} items/s" if rc == 0 else f"(diag rc {rc})"
    if os.environ.get("KINDS_PAIR_PLACEMENTS"):
        k = int(os.environ["KINDS_PAIR_PLACEMENTS"])
        arr = (C.c_double * k)()
        rc = ctx.lib.fgpu_diag_pair_placements(ctx.h, k, 1 << 26, arr)
        rates += "   other pair allocations, mixed: " + (" ".join(f"{x:.3e}" for x in arr) if rc == 0 else f"(rc {rc})")
    ptr, _ = ctx.bloom_devptr(0)
    free1, _ = torch.cuda.mem_get_info(0)
    print(f"{tag}: load_mark per pass {' '.join(f'{x:7.2f}' for x in out)} ms   {rates}   bloo1 at {ptr:#x}, device memory in use before / with the context {(total - free0) / 2**30:.1f} / {(total - free1) / 2**30:.1f} GiB", flush=True)
    ctx.close()


def child(mode, n):
    import torch
    import bench
    from faucet_amd import synth_det as sd
    dev = torch.device("cuda", 0)
    g = sd.make_genome(400_000_000, 4, dev)
    reads = sd.make_reads(g, 25_000_000, 100, 0.01, 4000, dev)
    del g
    torch.cuda.empty_cache()
    batches = bench.device_batches(reads, bench.batch_bounds(25_000_000, 2_500_000, 2))
    for i in range(n):
        one_context(reads, batches, f"pid {os.getpid()} context {i}")


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "procs"
    if mode == "--child":
        child(sys.argv[2], int(sys.argv[3]))
    else:
        n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
        if mode == "procs":
            for i in range(n):
                subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "procs", "1"], cwd=ROOT)
        elif mode == "contexts":
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "contexts", str(n)], cwd=ROOT)
        elif mode == "order":
            for i in range(n):
                subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "order", "1"], cwd=ROOT, env=dict(os.environ, FGPU_DEBUG_ALLOC_FIRST="1"))
