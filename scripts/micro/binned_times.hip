// Measurement aid, not part of the library (round 5): first-set times posted THROUGH BINS.  k_load_mark on the large filters is bound by its
// atomicMin into first[] (32 GiB for 2^33 bits: ~45 ps apiece, against ~23 ps while the table fits the 256 MiB Infinity Cache,
// profiles/r04_first_table_counters.txt).  atomicMin commutes, so the (bit, time) pairs of a batch may be applied in any order -- e.g. slice by slice
// of first[], each slice small enough for the cache:
//   A  direct   : n random atomicMin into the whole table (what the marking kernel does today)
//   B  binned   : (1) the pairs written to bins by the bit's high part, staged through LDS so that a bin receives runs of whole lines;
//                 (2) bin after bin, all of the device on ONE slice at a time: stream the bin's pairs, atomicMin.
// The table afterwards is the same in both (checked by a checksum).  What is printed: ms per 2^lgn pairs and ps per pair for A, B(1), B(2).
// usage: binned_times <log2 table entries = 33> <log2 pairs per launch = 28> <log2 entries per slice = 25 (128 MiB)>
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

__global__ void __launch_bounds__(256) k_direct(uint32_t* first, uint64_t mask, uint64_t n, uint64_t salt) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        atomicMin(first + (mix(i ^ salt) & mask), (uint32_t)(i >> 2));
}

// (1) one workgroup takes CHUNK pairs: histogram of their bins in LDS, one reservation per (workgroup, bin) in the bin's global cursor, then the pairs
// go to their places -- a bin receives CHUNK / n_bins pairs of this workgroup side by side (16384 / 256 = 64 pairs = 512 bytes)
constexpr int CHUNK = 16384;
__global__ void __launch_bounds__(256) k_bin(uint2* bins, unsigned long long* cursor, uint64_t bin_cap, uint32_t n_bins, int slice_lg, uint64_t mask, uint64_t n, uint64_t salt) {
    extern __shared__ uint32_t s[];          // [n_bins] counts, then [n_bins] bases (low 32 bits are enough inside one launch of < 2^32 pairs per bin)
    uint32_t* cnt = s;
    unsigned long long* base = (unsigned long long*)(s + n_bins);
    for (uint64_t c0 = (uint64_t)blockIdx.x * CHUNK; c0 < n; c0 += (uint64_t)gridDim.x * CHUNK) {
        for (uint32_t b = threadIdx.x; b < n_bins; b += 256) cnt[b] = 0;
        __syncthreads();
        uint32_t rank[CHUNK / 256];
#pragma unroll
        for (int q = 0; q < CHUNK / 256; q++) {
            const uint64_t i = c0 + (uint64_t)q * 256 + threadIdx.x;
            rank[q] = i < n ? atomicAdd(&cnt[(mix(i ^ salt) & mask) >> slice_lg], 1u) : 0u;
        }
        __syncthreads();
        for (uint32_t b = threadIdx.x; b < n_bins; b += 256) base[b] = cnt[b] ? atomicAdd(&cursor[b], (unsigned long long)cnt[b]) : 0ULL;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < CHUNK / 256; q++) {
            const uint64_t i = c0 + (uint64_t)q * 256 + threadIdx.x;
            if (i < n) {
                const uint64_t h = mix(i ^ salt) & mask;
                const uint32_t b = (uint32_t)(h >> slice_lg);
                bins[(uint64_t)b * bin_cap + base[b] + rank[q]] = make_uint2((uint32_t)(h & ((1ULL << slice_lg) - 1)), (uint32_t)(i >> 2));
            }
        }
        __syncthreads();
    }
}

// (2) one slice: its pairs streamed, atomicMin into the slice
__global__ void __launch_bounds__(256) k_apply(uint32_t* slice, const uint2* pairs, const unsigned long long* count) {
    const uint64_t n = *count;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint2 p = pairs[i];
        atomicMin(slice + p.x, p.y);
    }
}

__global__ void __launch_bounds__(256) k_checksum(const uint32_t* t, uint64_t n, unsigned long long* out) {
    unsigned long long acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) acc += (unsigned long long)t[i] * (i | 1);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}

int main(int argc, char** argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 33, lgn = argc > 2 ? atoi(argv[2]) : 28, slice_lg = argc > 3 ? atoi(argv[3]) : 25;
    const uint64_t entries = 1ULL << lg, n = 1ULL << lgn, mask = entries - 1;
    const uint32_t n_bins = (uint32_t)(entries >> slice_lg);
    const uint64_t bin_cap = (n / n_bins) * 5 / 4 + 65536;
    uint32_t* first; uint2* bins; unsigned long long *cursor, *sum;
    if (hipMalloc(&first, entries * 4) || hipMalloc(&bins, (uint64_t)n_bins * bin_cap * 8) || hipMalloc(&cursor, n_bins * 8) || hipMalloc(&sum, 8)) { printf("alloc failed\n"); return 1; }
    hipEvent_t e[4]; for (auto& x : e) hipEventCreate(&x);
    auto ms = [&](int a, int b) { float t; hipEventElapsedTime(&t, e[a], e[b]); return (double)t; };
    printf("table 2^%d entries (%.0f GiB), 2^%d pairs per launch, slices of 2^%d entries (%.0f MiB): %u bins\n", lg, entries * 4.0 / (1 << 30), lgn, slice_lg, (1ULL << slice_lg) * 4.0 / (1 << 20), n_bins);
    unsigned long long ref = 0;
    for (int rep = 0; rep < 3; rep++) {
        hipMemset(first, 0xFF, entries * 4); hipMemset(sum, 0, 8);
        hipDeviceSynchronize();
        hipEventRecord(e[0], 0);
        k_direct<<<4096, 256>>>(first, mask, n, 1234 + rep);
        hipEventRecord(e[1], 0);
        hipDeviceSynchronize();
        k_checksum<<<4096, 256>>>(first, entries, sum);
        hipMemcpy(&ref, sum, 8, hipMemcpyDeviceToHost);
        printf("A direct: %8.2f ms = %5.1f ps per pair\n", ms(0, 1), ms(0, 1) * 1e9 / (double)n);
        hipMemset(first, 0xFF, entries * 4); hipMemset(sum, 0, 8); hipMemset(cursor, 0, n_bins * 8);
        hipDeviceSynchronize();
        hipEventRecord(e[0], 0);
        k_bin<<<2048, 256, n_bins * 4 + n_bins * 8>>>(bins, cursor, bin_cap, n_bins, slice_lg, mask, n, 1234 + rep);
        hipEventRecord(e[1], 0);
        for (uint32_t b = 0; b < n_bins; b++) k_apply<<<2048, 256>>>(first + ((uint64_t)b << slice_lg), bins + (uint64_t)b * bin_cap, cursor + b);
        hipEventRecord(e[2], 0);
        hipDeviceSynchronize();
        std::vector<unsigned long long> c(n_bins);
        hipMemcpy(c.data(), cursor, n_bins * 8, hipMemcpyDeviceToHost);
        unsigned long long mx = 0; for (auto v : c) mx = v > mx ? v : mx;
        if (mx > bin_cap) { printf("a bin overflowed (%llu > %llu)\n", mx, (unsigned long long)bin_cap); return 1; }
        k_checksum<<<4096, 256>>>(first, entries, sum);
        unsigned long long got = 0;
        hipMemcpy(&got, sum, 8, hipMemcpyDeviceToHost);
        printf("B binned: (1) %7.2f ms = %5.1f ps, (2) %7.2f ms = %5.1f ps, together %7.2f ms = %5.1f ps per pair   table %s\n", ms(0, 1), ms(0, 1) * 1e9 / (double)n,
               ms(1, 2), ms(1, 2) * 1e9 / (double)n, ms(0, 2), ms(0, 2) * 1e9 / (double)n, got == ref ? "equal" : "DIFFERS");
    }
    return 0;
}
