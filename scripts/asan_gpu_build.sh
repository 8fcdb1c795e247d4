#!/bin/bash
# libfaucet_gpu.so with its HOST code under AddressSanitizer (the device side is not instrumented: -Xarch_host; -O1), next to the product library:
#   faucet_amd/build_asan/libfaucet_gpu_asan.so   (git-ignored, travels to the GPU box)
# Run (GPU box; the sanitizer's runtime has to be the first library of the process):
#   LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 \
#     FAUCET_GPU_LIB=faucet_amd/build_asan/libfaucet_gpu_asan.so python3 scripts/fuzz_shards.py 5000 5160
set -e
cd "$(dirname "$0")/.."
OUT=faucet_amd/build_asan
mkdir -p $OUT
FLAGS="-O1 -g -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-result -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer"
pids=()
for s in api pack load scan_pure scan_walk diag text stage3 pairs group; do
  /opt/rocm/bin/hipcc $FLAGS -x hip -c faucet_amd/csrc/$s.hip -o $OUT/$s.o &
  pids+=($!)
done
/opt/rocm/bin/hipcc $FLAGS -c faucet_amd/csrc/sizing.cpp -o $OUT/sizing.o
for p in "${pids[@]}"; do wait $p; done
# linked WITHOUT a sanitizer runtime: ROCm's clang runtime intercepts hsa_amd_memory_pool_allocate and dies at the first device allocation on
# this pool (no xnack); gcc's libasan has every __asan_* entry point clang's instrumentation calls and leaves HSA alone -- it is preloaded
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libfaucet_gpu_asan.so $OUT/*.o
ls -la $OUT/libfaucet_gpu_asan.so
