// Measurement aid, not part of the library (round 5): the WHOLE cost of the wide filter layout for large filters, kernel by kernel, beside
// today's -- scripts/micro/mark_model.hip (round 4) priced the marking kernel only.
//   A   today: 3 random 8-byte loads of {bloo1, bloo2}; a new k-mer posts 3 atomicMin into first[] (4 bytes per filter bit); then the resolve
//       kernel: one first[] load per new k-mer (its early exit)
//   Q   wide: 3 random 16-byte loads of {bloo1, bloo2, touched, touched twice}; a new k-mer ORs its bits into `touched` with the RETURNING
//       atomic (who was first?); bits seen touched already: `touched twice` + atomicMin into first[]; then the pull kernel: every new k-mer looks
//       at `touched twice` of its three words again (a first toucher must learn whether anybody came after it); then the sweep of the table
//       (bloo1 |= touched, both planes cleared): once per batch
//   N   lower bound: 16-byte loads + fire-and-forget atomicOr on the line just loaded, nothing else
// usage: mark_model2 <log2 filter bits> <share of new k-mers, per mille> [log2 k-mers per launch = 27]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
struct Item { uint64_t hA, hB; bool is_new; };
__device__ __forceinline__ Item item(uint64_t i, uint64_t salt, uint64_t bit_mask, uint32_t pm) {
    const uint64_t r = mix(i ^ salt);
    Item it;
    it.hA = r & bit_mask;
    it.hB = mix(r) | 1;
    it.is_new = (uint32_t)(mix(r + 7) % 1000) < pm;
    return it;
}

__global__ void __launch_bounds__(256) k_mark_a(uint2* pair8, uint32_t* first, uint64_t bit_mask, uint64_t n, uint32_t pm, uint64_t salt, unsigned long long* sink) {
    unsigned long long acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const Item it = item(i, salt, bit_mask, pm);
        uint64_t h = it.hA;
        uint2 v[3];
        for (int q = 0; q < 3; q++) { v[q] = pair8[h >> 5]; h = (h + it.hB) & bit_mask; }
        for (int q = 0; q < 3; q++) acc += v[q].x ^ v[q].y;
        if (it.is_new) {
            h = it.hA;
            for (int q = 0; q < 3; q++) { atomicMin(&first[h], (uint32_t)i); h = (h + it.hB) & bit_mask; }
        }
    }
    if (acc == 0x123456789ULL) *sink = acc;
}
__global__ void __launch_bounds__(256) k_resolve_a(const uint32_t* first, uint64_t bit_mask, uint64_t n, uint32_t pm, uint64_t salt, unsigned long long* sink) {
    unsigned long long acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const Item it = item(i, salt, bit_mask, pm);
        if (it.is_new) acc += first[it.hA] < (uint32_t)i ? 1 : 0;
    }
    if (acc == 0x123456789ULL) *sink = acc;
}
template <int RETURNING>
__global__ void __launch_bounds__(256) k_mark_q(uint4* quad, uint32_t* first, uint64_t bit_mask, uint64_t n, uint32_t pm, uint64_t salt, unsigned long long* sink) {
    unsigned long long acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const Item it = item(i, salt, bit_mask, pm);
        uint64_t h = it.hA;
        uint4 v[3];
        for (int q = 0; q < 3; q++) { v[q] = quad[h >> 5]; h = (h + it.hB) & bit_mask; }
        for (int q = 0; q < 3; q++) acc += v[q].x ^ v[q].y;
        if (it.is_new) {
            h = it.hA;
            for (int q = 0; q < 3; q++) {
                const uint32_t bit = 1u << (h & 31);
                if (RETURNING) {
                    bool later = (v[q].z & bit) != 0;
                    if (!later) later = (atomicOr(&quad[h >> 5].z, bit) & bit) != 0;
                    if (later) {
                        if (!(v[q].w & bit)) atomicOr(&quad[h >> 5].w, bit);
                        atomicMin(&first[h], (uint32_t)i);
                    } else {
                        acc += 3;      // (the deferred flag)
                    }
                } else {
                    atomicOr(&quad[h >> 5].z, bit);
                }
                h = (h + it.hB) & bit_mask;
            }
        }
    }
    if (acc == 0x123456789ULL) *sink = acc;
}
__global__ void __launch_bounds__(256) k_pull_q(const uint4* quad, uint32_t* first, uint64_t bit_mask, uint64_t n, uint32_t pm, uint64_t salt, unsigned long long* sink) {
    unsigned long long acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const Item it = item(i, salt, bit_mask, pm);
        if (!it.is_new) continue;
        uint64_t h = it.hA;
        uint32_t w[3];
        for (int q = 0; q < 3; q++) { w[q] = quad[h >> 5].w; h = (h + it.hB) & bit_mask; }
        h = it.hA;
        for (int q = 0; q < 3; q++) {
            if (w[q] & (1u << (h & 31))) { atomicMin(&first[h], (uint32_t)i); acc++; }
            h = (h + it.hB) & bit_mask;
        }
    }
    if (acc == 0x123456789ULL) *sink = acc;
}
__global__ void __launch_bounds__(256) k_sweep_q(uint4* quad, uint64_t words) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (uint64_t)gridDim.x * blockDim.x) {
        uint4 v = quad[i];
        if (v.z | v.w) { v.x |= v.z; v.z = 0; v.w = 0; quad[i] = v; }
    }
}

int main(int argc, char** argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 33;
    const uint32_t pm = argc > 2 ? (uint32_t)atoi(argv[2]) : 630;
    const uint64_t n = 1ULL << (argc > 3 ? atoi(argv[3]) : 27);
    const uint64_t bits = 1ULL << lg, words = bits / 32;
    uint2* p8; uint4* p16; uint32_t* first; unsigned long long* sink;
    if (hipMalloc(&p8, words * 8) || hipMalloc(&p16, words * 16) || hipMalloc(&first, bits * 4) || hipMalloc(&sink, 8)) { printf("alloc failed\n"); return 1; }
    hipMemset(p8, 0, words * 8); hipMemset(p16, 0, words * 16); hipMemset(first, 0xFF, bits * 4);
    hipEvent_t e[8]; for (auto& x : e) hipEventCreate(&x);
    auto ms = [&](int a, int b) { float t = 0; hipEventElapsedTime(&t, e[a], e[b]); return t; };
    for (int rep = 0; rep < 3; rep++) {
        const uint64_t salt = 1234567ULL * (rep + 1);
        hipMemset(p16, 0, words * 16);
        hipDeviceSynchronize();
        hipEventRecord(e[0], 0);
        hipLaunchKernelGGL(k_mark_a, dim3(8192), dim3(256), 0, 0, p8, first, bits - 1, n, pm, salt, sink);
        hipEventRecord(e[1], 0);
        hipLaunchKernelGGL(k_resolve_a, dim3(8192), dim3(256), 0, 0, first, bits - 1, n, pm, salt, sink);
        hipEventRecord(e[2], 0);
        hipLaunchKernelGGL(k_mark_q<1>, dim3(8192), dim3(256), 0, 0, p16, first, bits - 1, n, pm, salt + 1, sink);
        hipEventRecord(e[3], 0);
        hipLaunchKernelGGL(k_pull_q, dim3(8192), dim3(256), 0, 0, p16, first, bits - 1, n, pm, salt + 1, sink);
        hipEventRecord(e[4], 0);
        hipLaunchKernelGGL(k_sweep_q, dim3(8192), dim3(256), 0, 0, p16, words);
        hipEventRecord(e[5], 0);
        hipLaunchKernelGGL(k_mark_q<0>, dim3(8192), dim3(256), 0, 0, p16, first, bits - 1, n, pm, salt + 2, sink);
        hipEventRecord(e[6], 0);
        hipEventSynchronize(e[6]);
        printf("filter 2^%d bits, %u per mille new, %llu k-mers: A mark %.3f + resolve %.3f = %.3f ms | Q mark %.3f + pull %.3f + sweep %.3f = %.3f ms | N mark (fire-and-forget atomicOr on the loaded line) %.3f ms\n",
               lg, pm, (unsigned long long)n, ms(0, 1), ms(1, 2), ms(0, 2), ms(2, 3), ms(3, 4), ms(4, 5), ms(2, 5), ms(5, 6));
    }
    return 0;
}
