// Measurement aid, not part of the library (round 6): a "posted this batch" bit NEXT TO the carry, so that the carry is current at
// every batch's start and first[] is never swept -- does the pass get cheaper on 2^33-bit filters?
//   A   today: 3 random 8-byte loads of {carry, bloo2}; a k-mer with a bit outside the (lagging) carry posts 3 atomicMin into first[] (63 % of the
//       k-mers on config 4: 27 % new + carry lag); resolve: one first[] load per such k-mer; sweep of first[] (32 GiB) when an epoch closes
//   P   posted: 3 random 16-byte loads of {carry, bloo2, posted, -}; a k-mer with a bit outside the carry (46 %: the carry is one batch old at most)
//       posts 3 atomicMin into first[] AND 3 fire-and-forget atomicOr into `posted` of the line it has just loaded; resolve as before; fold
//       between batches: carry |= posted, posted = 0 over the 16-byte words (streaming); first[] is never swept
// usage: mark_model4 <log2 filter bits = 33> <per mille new, A = 630> <per mille new, P = 460> [log2 k-mers per launch = 27]
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
    it.hA = r & bit_mask; it.hB = mix(r) | 1; it.is_new = (uint32_t)(mix(r + 7) % 1000) < pm;
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
__global__ void __launch_bounds__(256) k_mark_p(uint4* quad, uint32_t* first, uint64_t bit_mask, uint64_t n, uint32_t pm, uint64_t salt, unsigned long long* sink) {
    unsigned long long acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const Item it = item(i, salt, bit_mask, pm);
        uint64_t h = it.hA;
        uint4 v[3];
        for (int q = 0; q < 3; q++) { v[q] = quad[h >> 5]; h = (h + it.hB) & bit_mask; }
        for (int q = 0; q < 3; q++) acc += v[q].x ^ v[q].y ^ v[q].z;
        if (it.is_new) {
            h = it.hA;
            for (int q = 0; q < 3; q++) {
                atomicMin(&first[h], (uint32_t)i);
                atomicOr(&((uint32_t*)&quad[h >> 5])[2], 1u << (h & 31));
                h = (h + it.hB) & bit_mask;
            }
        }
    }
    if (acc == 0x123456789ULL) *sink = acc;
}
__global__ void __launch_bounds__(256) k_resolve(const uint32_t* first, uint64_t bit_mask, uint64_t n, uint32_t pm, uint64_t salt, unsigned long long* sink) {
    unsigned long long acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const Item it = item(i, salt, bit_mask, pm);
        if (it.is_new) acc += first[it.hA];
    }
    if (acc == 0x123456789ULL) *sink = acc;
}
__global__ void __launch_bounds__(256) k_fold(uint4* quad, uint64_t words) {
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < words; w += (uint64_t)gridDim.x * blockDim.x) {
        uint4 v = quad[w];
        if (v.z) { v.x |= v.z; v.z = 0; quad[w] = v; }
    }
}
__global__ void __launch_bounds__(256) k_sweep(uint2* pair8, const uint4* first4, uint64_t words) {   // carry |= (first[bit] != never), 32 bits per filter word
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < words; w += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t m = 0;
        for (int q = 0; q < 8; q++) {
            const uint4 t = first4[w * 8 + q];
            m |= (t.x != 0xFFFFFFFFu ? 1u : 0u) << (4 * q) | (t.y != 0xFFFFFFFFu ? 2u : 0u) << (4 * q) | (t.z != 0xFFFFFFFFu ? 4u : 0u) << (4 * q) | (t.w != 0xFFFFFFFFu ? 8u : 0u) << (4 * q);
        }
        if (m) pair8[w].x |= m;
    }
}
int main(int argc, char** argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 33;
    const uint32_t pm_a = argc > 2 ? (uint32_t)atoi(argv[2]) : 630, pm_p = argc > 3 ? (uint32_t)atoi(argv[3]) : 460;
    const uint64_t n = 1ULL << (argc > 4 ? atoi(argv[4]) : 27);
    const uint64_t bits = 1ULL << lg, words = bits / 32;
    uint2* p8; uint4* q16; uint32_t* first; unsigned long long* sink;
    if (hipMalloc(&p8, words * 8) || hipMalloc(&q16, words * 16) || hipMalloc(&first, bits * 4) || hipMalloc(&sink, 8)) { printf("alloc failed\n"); return 1; }
    hipMemset(p8, 0, words * 8); hipMemset(q16, 0, words * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timed = [&](const char* name, auto launch, bool reset_first) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            if (reset_first) hipMemset(first, 0xFF, bits * 4);
            hipDeviceSynchronize();
            hipEventRecord(e0, 0); launch(rep); hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%-62s %8.2f ms\n", name, best);
        return best;
    };
    printf("2^%d filter bits, %llu k-mers per launch, new: A %u per mille, P %u per mille\n", lg, (unsigned long long)n, pm_a, pm_p);
    const float a_mark = timed("A  mark  (8-byte loads, atomicMin into first[])", [&](int r) { hipLaunchKernelGGL(k_mark_a, dim3(4096), dim3(256), 0, 0, p8, first, bits - 1, n, pm_a, 77ULL + r, sink); }, true);
    const float a_res = timed("A  resolve (one first[] load per new k-mer)", [&](int r) { hipLaunchKernelGGL(k_resolve, dim3(4096), dim3(256), 0, 0, first, bits - 1, n, pm_a, 77ULL + r, sink); }, false);
    const float a_sweep = timed("A  sweep of first[] (once per epoch, not per batch)", [&](int) { hipLaunchKernelGGL(k_sweep, dim3(4096), dim3(256), 0, 0, p8, (const uint4*)first, words); }, false);
    const float p_mark = timed("P  mark  (16-byte loads, atomicMin + atomicOr on the loaded line)", [&](int r) { hipLaunchKernelGGL(k_mark_p, dim3(4096), dim3(256), 0, 0, q16, first, bits - 1, n, pm_p, 77ULL + r, sink); }, true);
    const float p_res = timed("P  resolve", [&](int r) { hipLaunchKernelGGL(k_resolve, dim3(4096), dim3(256), 0, 0, first, bits - 1, n, pm_p, 77ULL + r, sink); }, false);
    const float p_fold = timed("P  fold  (carry |= posted, posted = 0; once per batch)", [&](int) { hipLaunchKernelGGL(k_fold, dim3(4096), dim3(256), 0, 0, q16, words); }, false);
    const float p_mark_same = timed("P  mark at A's share of new k-mers (what the layout alone costs)", [&](int r) { hipLaunchKernelGGL(k_mark_p, dim3(4096), dim3(256), 0, 0, q16, first, bits - 1, n, pm_a, 77ULL + r, sink); }, true);
    // config 4: 84 launches of ~1.67e8 k-mers per pass, 7 sweeps per pass (after batches 0, 1, 3, 7, 15, 31, 63 of the ramped schedule)
    const double per = 1.67e8 / (double)n;
    printf("per batch of 1.67e8 k-mers:  A  %.2f ms (mark %.2f + resolve %.2f + 7/84 sweep %.2f)   P  %.2f ms (mark %.2f + resolve %.2f + fold %.2f)   [layout alone: mark %.2f]\n",
           a_mark * per + a_res * per + a_sweep * 7 / 84, a_mark * per, a_res * per, a_sweep * 7 / 84, p_mark * per + p_res * per + p_fold, p_mark * per, p_res * per, p_fold, p_mark_same * per);
    return 0;
}
