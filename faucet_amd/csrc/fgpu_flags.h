// fgpu_flags.h — testForJunction / JChecker::jcheck on the device (shared by the pure stage, which evaluates them where the
// walk is expected, and by the walk itself, which evaluates them on the spot where the expectation was wrong).
#pragma once
#include "fgpu_device.h"

namespace {

// JChecker::jcheck: depth-first search for one chain of j present extensions (same truth value as the
// reference's level-by-level search; nothing else about it is observable).
__device__ bool jcheck_dfs(uint64_t kmer, const FdParams& fp, const uint32_t* __restrict__ bloom) {
    if (fp.j == 0) return true;
    uint64_t stack_k[8];
    int stack_nt[8];
    int depth = 0;
    stack_k[0] = kmer;
    stack_nt[0] = 0;
    const int J = fp.j < 8 ? fp.j : 8;
    while (depth >= 0) {
        if (stack_nt[depth] == 4) { depth--; continue; }
        int nt = stack_nt[depth]++;
        uint64_t e = ((stack_k[depth] << 2) | (uint64_t)nt) & fp.kmask;
        if (fd_bloom_contains_canon_lazy(bloom, fd_canon(e, fp.k), fp.tai_mask, fp.n_hash)) {
            if (depth + 1 == J) return true;
            depth++;
            stack_k[depth] = e;
            stack_nt[depth] = 0;
        }
    }
    return false;
}

// testForJunction for the k-mer `key` (already oriented towards the extension) with real next base `real`
__device__ __forceinline__ void test_for_junction(uint64_t key, int real, const FdParams& fp, const uint32_t* __restrict__ bloom,
                                                  bool& flag, int& njc) {
    flag = false;
    njc = 0;
    for (int nt = 0; nt < 4; nt++) {
        if (nt == real) continue;
        uint64_t e = ((key << 2) | (uint64_t)nt) & fp.kmask;
        if (fd_bloom_contains_canon_lazy(bloom, fd_canon(e, fp.k), fp.tai_mask, fp.n_hash)) {
            njc++;
            if (jcheck_dfs(e, fp, bloom)) { flag = true; return; }
        }
    }
}

// position of the n-th set bit of x (n < popcount(x))
__device__ __forceinline__ int select_bit(uint64_t x, int n) {
    int pos = 0;
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) {
        const int c = __popcll(x & ((1ULL << sh) - 1));
        if (n >= c) { x >>= sh; pos += sh; n -= c; }
    }
    return pos;
}

}  // namespace
