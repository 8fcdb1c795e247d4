// Measurement aid, not part of the library: what k_load_mark's access pattern costs in two table layouts.
//   A (today):    per k-mer 3 random 8-byte loads from the interleaved {bloo1, bloo2} words; a "new" k-mer posts 3 atomicMin into a table of
//                 first-set times (4 bytes per filter bit) at the same bit indices
//   B (proposed): per k-mer 3 random 16-byte loads from {bloo1, bloo2, touched-this-batch, touched-twice} words; a new k-mer sets its bit in the
//                 third field of the word it has just loaded (atomicOr on the same line), no time table
// usage: mark_model <log2 filter bits> <share of new k-mers, per mille> [items per launch, log2 = 27]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

template <int VARIANT>
__global__ void __launch_bounds__(256) k_model(uint2* pair8, uint4* pair16, uint32_t* first, uint64_t bit_mask, uint64_t n, uint32_t new_permille,
                                               uint64_t salt, unsigned long long* sink) {
    unsigned long long acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = mix(i ^ salt);
        const uint64_t hA = r & bit_mask, hB = mix(r) | 1;
        const bool is_new = (uint32_t)(mix(r + 7) % 1000) < new_permille;
        uint64_t h = hA;
        uint32_t seen = 0;
        if (VARIANT == 0) {
            uint2 v[3];
            for (int q = 0; q < 3; q++) { v[q] = pair8[h >> 5]; h = (h + hB) & bit_mask; }
            for (int q = 0; q < 3; q++) seen += v[q].x ^ v[q].y;
            if (is_new) {
                h = hA;
                for (int q = 0; q < 3; q++) { atomicMin(&first[h], (uint32_t)i); h = (h + hB) & bit_mask; }
            }
        } else {
            uint4 v[3];
            for (int q = 0; q < 3; q++) { v[q] = pair16[h >> 5]; h = (h + hB) & bit_mask; }
            for (int q = 0; q < 3; q++) seen += v[q].x ^ v[q].y ^ v[q].z;
            if (is_new) {
                h = hA;
                for (int q = 0; q < 3; q++) {
                    const uint32_t old = atomicOr(&pair16[h >> 5].z, 1u << (h & 31));
                    if (old & (1u << (h & 31))) atomicOr(&pair16[h >> 5].w, 1u << (h & 31));
                    h = (h + hB) & bit_mask;
                }
            }
        }
        acc += seen;
    }
    if (acc == 0x123456789ULL) *sink = acc;
}

int main(int argc, char** argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 33;
    const uint32_t pm = argc > 2 ? (uint32_t)atoi(argv[2]) : 630;
    const uint64_t n = 1ULL << (argc > 3 ? atoi(argv[3]) : 27);
    const uint64_t bits = 1ULL << lg, words = bits / 32;
    uint2* p8; uint4* p16; uint32_t* first; unsigned long long* sink;
    if (hipMalloc(&p8, words * 8) || hipMalloc(&p16, words * 16) || hipMalloc(&first, bits * 4) || hipMalloc(&sink, 8)) { printf("alloc failed\n"); return 1; }
    hipMemset(p8, 0, words * 8); hipMemset(p16, 0, words * 16); hipMemset(first, 0xFF, bits * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int variant = 0; variant < 2; variant++)
        for (int rep = 0; rep < 3; rep++) {
            hipMemset(p16, 0, words * 16);
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            if (variant == 0) hipLaunchKernelGGL(k_model<0>, dim3(8192), dim3(256), 0, 0, p8, p16, first, bits - 1, n, pm, 1234567ULL * (rep + 1), sink);
            else hipLaunchKernelGGL(k_model<1>, dim3(8192), dim3(256), 0, 0, p8, p16, first, bits - 1, n, pm, 1234567ULL * (rep + 1), sink);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            printf("filter 2^%d bits, %u per mille new, %s: %.3f ms for %llu k-mers = %.3g k-mers/s\n", lg, pm, variant ? "B (16-byte words, same-line atomicOr)" : "A (8-byte words + first[] atomicMin)",
                   ms, (unsigned long long)n, n / (ms * 1e-3));
        }
    return 0;
}
