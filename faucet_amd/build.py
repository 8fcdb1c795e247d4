"""Build libfaucet_gpu.so (HIP, gfx950 only) and the host `faucet` CLI in-tree.

    python -m faucet_amd.build            # incremental
    python -m faucet_amd.build --force

hipcc cross-compiles for gfx950 without a GPU; the built .so travels to the GPU box with the tree
(it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libfaucet_gpu.so")
CLI = os.path.join(HERE, "faucet")
HIP_SOURCES = ["api.hip", "pack.hip", "load.hip", "scan_pure.hip", "scan_walk.hip", "diag.hip", "text.hip", "stage3.hip", "pairs.hip", "group.hip"]
CPP_SOURCES = ["sizing.cpp"]
HEADERS = ["fgpu_ctx.h", "fgpu_device.h", "fgpu_flags.h", os.path.join(ROOT, "include", "faucet_gpu.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-result", "-Wno-unused-value"] + os.environ.get("FGPU_EXTRA_CXXFLAGS", "").split()


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + r.stderr)
        raise RuntimeError("build failed: " + os.path.basename(cmd[-1]))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)


def build(force: bool = False, jobs: int = 6) -> str:
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    objs, procs = [], []
    for src in HIP_SOURCES + CPP_SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(objdir, src + ".o")
        objs.append(obj)
        if force or _newer(obj, [sp] + hdrs):
            cmd = [HIPCC] + FLAGS + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", sp, "-o", obj]
            procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
            if len(procs) >= jobs:
                _wait(procs)
    _wait(procs)
    if force or _newer(LIB, objs):
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    main = os.path.join(HERE, "host", "faucet_main.cpp")
    if os.path.exists(main) and (force or _newer(CLI, [main, LIB] + [os.path.join(HERE, "host", h) for h in ("junction_order.h", "shard_host.h", "text_source.h", "pair_loop.h")] + hdrs)):
        _run(["g++", "-std=c++11", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), main, "-o", CLI,
              "-L", HERE, "-lfaucet_gpu", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link," + "/opt/rocm/lib", "-lpthread"])
    return LIB


def _wait(procs):
    failed = None
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(" ".join(cmd) + "\n" + out)
            failed = cmd
        elif out.strip():
            sys.stderr.write(out)
    procs.clear()
    if failed:
        raise RuntimeError("build failed: " + os.path.basename(failed[-3]))


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
