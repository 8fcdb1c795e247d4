// diag.hip — measured ceilings of the device for the two access patterns of the path:
//   * a streaming copy (what pack / the bit planes see), and
//   * independent random 32-bit accesses into a table of a given size (what every Bloom probe, the first-set-time
//     array of the load pass and the junction presence filter see).  A random access moves one 64-byte sector at
//     least, so accesses/s x 64 B is the "random-64 B-gather" rate the roofline fractions are put next to.
// Nothing of the product path calls these; bench.py reports them beside the k-mer rates.
#include "fgpu_ctx.h"

namespace {

__global__ void __launch_bounds__(256) k_diag_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, uint64_t n16) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}

__device__ __forceinline__ uint64_t diag_rand(uint64_t x) {
    x *= 0x9E3779B97F4A7C15ULL;
    x ^= x >> 29;
    x *= 0xBF58476D1CE4E5B9ULL;
    x ^= x >> 32;
    return x;
}

// mode 0: load, 1: atomicMin, 2: test then atomicOr (the filter's add).  8 independent accesses per thread per round.
template <int MODE>
__global__ void __launch_bounds__(256) k_diag_random(uint32_t* __restrict__ table, uint64_t word_mask, uint64_t n_rounds, uint64_t salt,
                                                     uint32_t* __restrict__ sink) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rounds; r += stride) {
        uint64_t idx[8];
#pragma unroll
        for (int u = 0; u < 8; u++) idx[u] = diag_rand((r * 8 + u) ^ salt) & word_mask;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (MODE == 0) acc ^= table[idx[u]];
            else if (MODE == 1) atomicMin(&table[idx[u]], (uint32_t)(r + u));
            else {
                const uint32_t bit = 1u << (u + (r & 15));
                if (!(table[idx[u]] & bit)) atomicOr(&table[idx[u]], bit);
            }
        }
    }
    if (MODE == 0 && acc == 0x12345678u) sink[0] = acc;   // keeps the loads alive
}

}  // namespace

extern "C" int fgpu_diag_stream_copy(fgpu_ctx* ctx, uint64_t bytes, int iters, double* gb_per_s) {
    if (!ctx || !gb_per_s || bytes < 4096 || iters < 1) return FGPU_ERR_ARG;
    bytes &= ~(uint64_t)15;
    void *src = nullptr, *dst = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    FGPU_HIP(hipMalloc(&src, bytes));
    if (hipMalloc(&dst, bytes) != hipSuccess) { hipFree(src); ctx->err = "diag: out of device memory"; return FGPU_ERR_NOMEM; }
    hipMemsetAsync(src, 1, bytes, ctx->stream);
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const uint64_t n16 = bytes / 16;
    for (int i = -1; i < iters; i++) {   // one untimed pass first
        if (i == 0) hipEventRecord(e0, ctx->stream);
        hipLaunchKernelGGL(k_diag_copy, dim3(fgpu_grid(n16, 256)), dim3(256), 0, ctx->stream, (const uint4*)src, (uint4*)dst, n16);
    }
    hipEventRecord(e1, ctx->stream);
    hipError_t e = hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipFree(src);
    hipFree(dst);
    FGPU_HIP(e);
    *gb_per_s = 2.0 * (double)bytes * iters / (ms * 1e-3) / 1e9;   // read + write
    return FGPU_OK;
}

extern "C" int fgpu_diag_random_access(fgpu_ctx* ctx, uint64_t table_bytes, uint64_t n_access, int mode, int iters, double* access_per_s) {
    if (!ctx || !access_per_s || table_bytes < 4096 || (table_bytes & (table_bytes - 1)) || n_access < 8 || mode < 0 || mode > 2 || iters < 1)
        return FGPU_ERR_ARG;
    uint32_t* table = nullptr;
    uint32_t* sink = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    FGPU_HIP(hipMalloc(&table, table_bytes));
    if (hipMalloc(&sink, 64) != hipSuccess) { hipFree(table); ctx->err = "diag: out of device memory"; return FGPU_ERR_NOMEM; }
    hipMemsetAsync(table, mode == 1 ? 0xff : 0, table_bytes, ctx->stream);
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const uint64_t rounds = n_access / 8, mask = table_bytes / 4 - 1;
    const unsigned grid = fgpu_grid(rounds, 256);
    for (int i = -1; i < iters; i++) {
        if (i == 0) hipEventRecord(e0, ctx->stream);
        const uint64_t salt = 0x5851F42D4C957F2DULL * (uint64_t)(i + 2);
        if (mode == 0) hipLaunchKernelGGL(k_diag_random<0>, dim3(grid), dim3(256), 0, ctx->stream, table, mask, rounds, salt, sink);
        else if (mode == 1) hipLaunchKernelGGL(k_diag_random<1>, dim3(grid), dim3(256), 0, ctx->stream, table, mask, rounds, salt, sink);
        else hipLaunchKernelGGL(k_diag_random<2>, dim3(grid), dim3(256), 0, ctx->stream, table, mask, rounds, salt, sink);
    }
    hipEventRecord(e1, ctx->stream);
    hipError_t e = hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipFree(table);
    hipFree(sink);
    FGPU_HIP(e);
    *access_per_s = (double)(rounds * 8) * iters / (ms * 1e-3);
    return FGPU_OK;
}

// what the runtime reports for the memory system of the context's device (printed beside the measured ceilings)
extern "C" int fgpu_diag_device_attr(fgpu_ctx* ctx, int32_t* mem_clock_khz, int32_t* mem_bus_bits, int32_t* l2_bytes, int32_t* compute_units) {
    if (!ctx || !mem_clock_khz || !mem_bus_bits || !l2_bytes || !compute_units) return FGPU_ERR_ARG;
    int v = 0;
    FGPU_HIP(hipDeviceGetAttribute(&v, hipDeviceAttributeMemoryClockRate, ctx->prm.device));
    *mem_clock_khz = v;
    FGPU_HIP(hipDeviceGetAttribute(&v, hipDeviceAttributeMemoryBusWidth, ctx->prm.device));
    *mem_bus_bits = v;
    FGPU_HIP(hipDeviceGetAttribute(&v, hipDeviceAttributeL2CacheSize, ctx->prm.device));
    *l2_bytes = v;
    FGPU_HIP(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, ctx->prm.device));
    *compute_units = v;
    return FGPU_OK;
}
