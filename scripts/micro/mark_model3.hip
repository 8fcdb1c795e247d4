// Measurement aid, not part of the library (round 5): does the first-set time of a bit cost less when it lies NEXT TO the filter word the marking
// kernel has just loaded?  (scripts/micro/mark_model2.hip: a fire-and-forget atomic on the loaded line costs half of an atomicMin into the
// separate table of times.)  Same algorithm as today in every variant -- 3 random loads of {bloo1, bloo2}, a new k-mer posts 3 atomicMin -- only
// the address of the time changes:
//   A    today: times in their own array, first[bit]
//   R192 one 192-byte record per 32 filter bits: {bloo1, bloo2, 56 bytes unused} then the 32 times (two more lines of the same record)
//   R256 the same, records of 256 bytes (aligned: a record never straddles a 256-byte boundary)
//   R160 {bloo1, bloo2, 24 bytes unused} + 32 times = 160 bytes
// usage: mark_model3 <log2 filter bits> <share of new k-mers, per mille> [log2 k-mers per launch = 27]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

// rec_words: 32-bit words per record (0 = separate arrays); time_at: word offset of the times inside a record
__global__ void __launch_bounds__(256) k_mark(uint32_t* table, uint32_t* first, uint32_t rec_words, uint32_t time_at, uint64_t bit_mask, uint64_t n, uint32_t pm,
                                              uint64_t salt, unsigned long long* sink) {
    unsigned long long acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = mix(i ^ salt);
        const uint64_t hA = r & bit_mask, hB = mix(r) | 1;
        const bool is_new = (uint32_t)(mix(r + 7) % 1000) < pm;
        uint64_t h = hA;
        uint2 v[3];
        for (int q = 0; q < 3; q++) {
            v[q] = rec_words ? *(const uint2*)(table + (h >> 5) * rec_words) : ((const uint2*)table)[h >> 5];
            h = (h + hB) & bit_mask;
        }
        for (int q = 0; q < 3; q++) acc += v[q].x ^ v[q].y;
        if (is_new) {
            h = hA;
            for (int q = 0; q < 3; q++) {
                uint32_t* t = rec_words ? table + (h >> 5) * rec_words + time_at + (h & 31) : first + h;
                atomicMin(t, (uint32_t)i);
                h = (h + hB) & bit_mask;
            }
        }
    }
    if (acc == 0x123456789ULL) *sink = acc;
}

int main(int argc, char** argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 33;
    const uint32_t pm = argc > 2 ? (uint32_t)atoi(argv[2]) : 630;
    const uint64_t n = 1ULL << (argc > 3 ? atoi(argv[3]) : 27);
    const uint64_t bits = 1ULL << lg, words = bits / 32;
    uint32_t *p8, *first, *rec; unsigned long long* sink;
    if (hipMalloc(&p8, words * 8) || hipMalloc(&first, bits * 4) || hipMalloc(&rec, words * 256) || hipMalloc(&sink, 8)) { printf("alloc failed\n"); return 1; }
    hipMemset(p8, 0, words * 8); hipMemset(first, 0xFF, bits * 4); hipMemset(rec, 0xFF, words * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct V { const char* name; uint32_t rec_words, time_at; } vs[] = {{"A (separate arrays)", 0, 0}, {"R192 (192-byte records)", 48, 16}, {"R256 (256-byte records)", 64, 16}, {"R160 (160-byte records)", 40, 8}};
    for (const V& v : vs)
        for (int rep = 0; rep < 3; rep++) {
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_mark, dim3(8192), dim3(256), 0, 0, v.rec_words ? rec : p8, first, v.rec_words, v.time_at, bits - 1, n, pm, 1234567ULL * (rep + 1), sink);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            printf("filter 2^%d bits, %u per mille new, %-28s %.3f ms for %llu k-mers\n", lg, pm, v.name, ms, (unsigned long long)n);
        }
    return 0;
}
