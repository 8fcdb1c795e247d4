#!/bin/bash
# scripts/fuzz_shards.py against libfaucet_gpu_asan.so (scripts/asan_gpu_build.sh): the library's HOST code under AddressSanitizer on the GPU box.
#   scripts/asan_gpu_fuzz.sh FIRST LAST [script]     (gcc's libasan + libstdc++ preloaded; torch's libraries on the path: libasan's dlopen drops RUNPATHs)
cd "$(dirname "$0")/.."
export LD_LIBRARY_PATH=/usr/local/lib/python3.10/dist-packages/torch/lib:$LD_LIBRARY_PATH
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libstdc++.so)"
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=1
export FAUCET_GPU_LIB=$PWD/faucet_amd/build_asan/libfaucet_gpu_asan.so
python3 ${3:-scripts/fuzz_shards.py} $1 $2
