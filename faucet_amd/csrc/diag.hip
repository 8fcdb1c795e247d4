// diag.hip — measured ceilings of the device for the two access patterns of the path:
//   * a streaming copy (what pack / the bit planes see), and
//   * independent random 32-bit accesses into a table of a given size (what every Bloom probe, the first-set-time
//     array of the load pass and the junction presence filter see).  A random access moves one 64-byte sector at
//     least, so accesses/s x 64 B is the "random-64 B-gather" rate the roofline fractions are put next to.
// Nothing of the product path calls these; bench.py reports them beside the k-mer rates.
#include <functional>
#include <vector>

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

// ---- NS1 ("LDS-staged per-wave query buckets ... 128 B-coalesced Bloom access"), built and measured -------------------------------------
// The filter's bit positions are the file format's, so the only blocking available is on the QUERY side: bin the probes by filter slice
// (one slice = one XCD's 4 MiB L2), then probe slice by slice from workgroups that run on the XCD whose L2 holds the slice.
//   k_diag_probe_direct   what every kernel of the path does today: address -> one random load -> answer bit (ballot, item order)
//   k_diag_bin            4096 probes per block: slice histogram in LDS, ONE global reservation per (block, slice), records
//                         {bit within slice, item} written to the slice's queue in block-contiguous runs
//   k_diag_probe_binned   blocks b, b+8, b+16 ... (one XCD: workgroups are dealt round-robin over the 8 XCDs) serve the slices
//                         s = b mod 8, +8, ...: queue records streamed, bits tested in the L2-resident slice, answers as ballot words in
//                         QUEUE order (what a next, equally binned, level of a probe chain would consume)
// fgpu_diag_binned_probes times both forms on the same probe addresses; scripts/binned_probe_ab.py adds the rocprofv3 counters.
constexpr int DIAG_BIN_PER_THREAD = 16;
constexpr int DIAG_MAX_SLICES = 512;

__global__ void __launch_bounds__(256) k_diag_probe_direct(const uint32_t* __restrict__ table, uint64_t bit_mask, uint64_t n, uint64_t salt,
                                                           uint64_t* __restrict__ answers) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {   // n is a multiple of 64
        const uint64_t a = diag_rand(i ^ salt) & bit_mask;
        const bool hit = (table[a >> 5] >> (a & 31)) & 1u;
        const uint64_t m = __ballot(hit);
        if (fd_lane() == 0) answers[i >> 6] = m;
    }
}

__global__ void __launch_bounds__(256) k_diag_bin(uint64_t bit_mask, int slice_shift, int n_slices, uint64_t n, uint64_t salt, uint64_t cap,
                                                  unsigned long long* __restrict__ qcount, uint2* __restrict__ queue) {
    __shared__ unsigned s_cnt[DIAG_MAX_SLICES], s_base[DIAG_MAX_SLICES];
    for (int q = threadIdx.x; q < n_slices; q += 256) s_cnt[q] = 0;
    __syncthreads();
    const uint64_t first = (uint64_t)blockIdx.x * 256 * DIAG_BIN_PER_THREAD;
    uint32_t addr[DIAG_BIN_PER_THREAD];
    unsigned rank[DIAG_BIN_PER_THREAD];
#pragma unroll
    for (int u = 0; u < DIAG_BIN_PER_THREAD; u++) {
        const uint64_t i = first + (uint64_t)u * 256 + threadIdx.x;
        addr[u] = (uint32_t)(diag_rand(i ^ salt) & bit_mask);
        rank[u] = i < n ? atomicAdd(&s_cnt[addr[u] >> slice_shift], 1u) : 0u;      // LDS: position within this block's run of the slice
    }
    __syncthreads();
    for (int q = threadIdx.x; q < n_slices; q += 256) s_base[q] = s_cnt[q] ? (unsigned)atomicAdd(&qcount[q], (unsigned long long)s_cnt[q]) : 0u;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < DIAG_BIN_PER_THREAD; u++) {
        const uint64_t i = first + (uint64_t)u * 256 + threadIdx.x;
        if (i >= n) continue;
        const uint32_t sl = addr[u] >> slice_shift;
        const uint64_t pos = (uint64_t)s_base[sl] + rank[u];
        if (pos < cap) queue[(uint64_t)sl * cap + pos] = make_uint2(addr[u], (uint32_t)i);
    }
}

__global__ void __launch_bounds__(256) k_diag_probe_binned(const uint32_t* __restrict__ table, int n_slices, uint64_t cap,
                                                           const unsigned long long* __restrict__ qcount, const uint2* __restrict__ queue,
                                                           uint64_t* __restrict__ answers) {
    const unsigned xcd = blockIdx.x & 7u, lane_block = blockIdx.x >> 3, blocks_per_xcd = gridDim.x >> 3;
    for (int sl = (int)xcd; sl < n_slices; sl += 8) {
        uint64_t cnt = qcount[sl];
        if (cnt > cap) cnt = cap;
        const uint64_t padded = (cnt + 63) & ~63ULL;
        const uint2* q = queue + (uint64_t)sl * cap;
        for (uint64_t i = (uint64_t)lane_block * 256 + threadIdx.x; i < padded; i += (uint64_t)blocks_per_xcd * 256) {
            bool hit = false;
            if (i < cnt) {
                const uint2 r = q[i];
                hit = (table[r.x >> 5] >> (r.x & 31)) & 1u;
            }
            const uint64_t m = __ballot(hit);
            if (fd_lane() == 0) answers[((uint64_t)sl * cap + i) >> 6] = m;
        }
    }
}


// ---- NS1, the whole CHAIN (round 3): Bloom::contains with its early exit -- n_hash dependent bit tests at (hA + i hB) mod tai -- for n items,
// (A) directly, one item per lane until its chain ends, and (B) with the FIRST level binned by filter slice: k_diag_bin queues {first bit,
// item}, k_diag_first_binned tests the first bits slice by slice from the XCD that holds the slice and hands the SURVIVORS back as a dense
// list of items (one reservation per wave: no random write, which is what a per-probe answer plane would cost), k_diag_chain_rest runs the
// rest of their chains directly and ORs the rare "all bits set" into the item-order answer plane.  This is the shape a binned first level
// of k_scan_flags_sm would have (survivors re-enter the state machine as a dense list, results are sparse flags).
__device__ __forceinline__ void diag_item_hashes(uint64_t i, uint64_t salt, uint64_t bit_mask, uint64_t& hA, uint64_t& hB) {
    hA = diag_rand(i ^ salt) & bit_mask;
    hB = diag_rand((i ^ salt) + 0x9E3779B97F4A7C15ULL) & bit_mask;
}

__global__ void __launch_bounds__(256) k_diag_chain_direct(const uint32_t* __restrict__ table, uint64_t bit_mask, int n_hash, uint64_t n, uint64_t salt,
                                                           uint64_t* __restrict__ answers) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {   // n is a multiple of 64
        uint64_t hA, hB;
        diag_item_hashes(i, salt, bit_mask, hA, hB);
        bool pass = true;
        uint64_t h = hA;
        for (int b = 0; b < n_hash; b++) {
            if (!((table[h >> 5] >> (h & 31)) & 1u)) { pass = false; break; }
            h = (h + hB) & bit_mask;
        }
        const uint64_t m = __ballot(pass);
        if (fd_lane() == 0) answers[i >> 6] = m;
    }
}

// first bits, slice by slice from one XCD each; survivors as a dense list of item numbers.  A block takes 4096 queue records at a time and
// reserves room for their survivors with ONE atomic (a reservation per wave was 4 M same-address atomics per pass: 42 of the kernel's 50 ms)
__global__ void __launch_bounds__(256) k_diag_first_binned(const uint32_t* __restrict__ table, int n_slices, uint64_t cap,
                                                           const unsigned long long* __restrict__ qcount, const uint2* __restrict__ queue,
                                                           uint32_t* __restrict__ survivors, unsigned long long* __restrict__ n_survivors) {
    __shared__ unsigned s_wave[4];
    __shared__ unsigned long long s_base;
    const unsigned xcd = blockIdx.x & 7u, lane_block = blockIdx.x >> 3, blocks_per_xcd = gridDim.x >> 3;
    const int wave = (int)(threadIdx.x >> 6);
    for (int sl = (int)xcd; sl < n_slices; sl += 8) {
        uint64_t cnt = qcount[sl];
        if (cnt > cap) cnt = cap;
        const uint2* q = queue + (uint64_t)sl * cap;
        for (uint64_t c0 = (uint64_t)lane_block * 4096; c0 < cnt; c0 += (uint64_t)blocks_per_xcd * 4096) {   // uniform trip count per block
            uint32_t item[DIAG_BIN_PER_THREAD];
            unsigned hits = 0;
#pragma unroll
            for (int u = 0; u < DIAG_BIN_PER_THREAD; u++) {
                const uint64_t i = c0 + (uint64_t)u * 256 + threadIdx.x;
                bool hit = false;
                if (i < cnt) {
                    const uint2 r = q[i];
                    item[u] = r.y;
                    hit = (table[r.x >> 5] >> (r.x & 31)) & 1u;
                }
                if (hit) hits |= 1u << u;
            }
            unsigned mine = (unsigned)__popc(hits), incl = mine;
            for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(incl, o, 64); if (fd_lane() >= o) incl += t; }
            if (fd_lane() == 63) s_wave[wave] = incl;
            __syncthreads();
            if (threadIdx.x == 0) {
                const unsigned total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
                s_base = total ? atomicAdd(n_survivors, (unsigned long long)total) : 0ULL;
            }
            __syncthreads();
            unsigned long long at = s_base + (incl - mine);
            for (int w = 0; w < wave; w++) at += s_wave[w];
#pragma unroll
            for (int u = 0; u < DIAG_BIN_PER_THREAD; u++)
                if (hits & (1u << u)) survivors[at++] = item[u];
            __syncthreads();      // s_wave / s_base are reused by the next round
        }
    }
}

// the rest of the survivors' chains, directly; "all bits set" is rare and goes to the item-order plane by atomicOr
__global__ void __launch_bounds__(256) k_diag_chain_rest(const uint32_t* __restrict__ table, uint64_t bit_mask, int n_hash, uint64_t salt,
                                                         const uint32_t* __restrict__ survivors, const unsigned long long* __restrict__ n_survivors,
                                                         unsigned long long* __restrict__ answers) {
    const uint64_t n = *n_survivors, stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n; s += stride) {
        const uint64_t i = survivors[s];
        uint64_t hA, hB;
        diag_item_hashes(i, salt, bit_mask, hA, hB);
        bool pass = true;
        uint64_t h = (hA + hB) & bit_mask;
        for (int b = 1; b < n_hash; b++) {
            if (!((table[h >> 5] >> (h & 31)) & 1u)) { pass = false; break; }
            h = (h + hB) & bit_mask;
        }
        if (pass) atomicOr(&answers[i >> 6], 1ULL << (i & 63));
    }
}

// queue records {first bit, item} of the chains' first level (k_diag_bin's scheme, the address is the item's hA)
__global__ void __launch_bounds__(256) k_diag_bin_items(uint64_t bit_mask, int slice_shift, int n_slices, uint64_t n, uint64_t salt, uint64_t cap,
                                                        unsigned long long* __restrict__ qcount, uint2* __restrict__ queue) {
    __shared__ unsigned s_cnt[DIAG_MAX_SLICES], s_base[DIAG_MAX_SLICES];
    for (int q = threadIdx.x; q < n_slices; q += 256) s_cnt[q] = 0;
    __syncthreads();
    const uint64_t first = (uint64_t)blockIdx.x * 256 * DIAG_BIN_PER_THREAD;
    uint32_t addr[DIAG_BIN_PER_THREAD];
    unsigned rank[DIAG_BIN_PER_THREAD];
#pragma unroll
    for (int u = 0; u < DIAG_BIN_PER_THREAD; u++) {
        const uint64_t i = first + (uint64_t)u * 256 + threadIdx.x;
        uint64_t hA, hB;
        diag_item_hashes(i, salt, bit_mask, hA, hB);
        addr[u] = (uint32_t)hA;
        rank[u] = i < n ? atomicAdd(&s_cnt[addr[u] >> slice_shift], 1u) : 0u;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < n_slices; q += 256) s_base[q] = s_cnt[q] ? (unsigned)atomicAdd(&qcount[q], (unsigned long long)s_cnt[q]) : 0u;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < DIAG_BIN_PER_THREAD; u++) {
        const uint64_t i = first + (uint64_t)u * 256 + threadIdx.x;
        if (i >= n) continue;
        const uint32_t sl = addr[u] >> slice_shift;
        const uint64_t pos = (uint64_t)s_base[sl] + rank[u];
        if (pos < cap) queue[(uint64_t)sl * cap + pos] = make_uint2(addr[u], (uint32_t)i);
    }
}

}  // namespace

extern "C" int fgpu_diag_binned_probes(fgpu_ctx* ctx, uint64_t table_bytes, uint64_t n_probes, uint64_t slice_bytes, int iters,
                                       double* direct_per_s, double* binned_per_s, double* bin_ms, double* probe_ms) {
    if (!ctx || !direct_per_s || !binned_per_s || !bin_ms || !probe_ms || iters < 1 || table_bytes < (1u << 16) || (table_bytes & (table_bytes - 1)) ||
        table_bytes > (1ULL << 29) || (slice_bytes & (slice_bytes - 1)) || slice_bytes < 4096 || slice_bytes > table_bytes ||
        table_bytes / slice_bytes > DIAG_MAX_SLICES || n_probes < (1u << 16))
        return FGPU_ERR_ARG;
    n_probes &= ~(uint64_t)(256 * DIAG_BIN_PER_THREAD - 1);
    const int n_slices = (int)(table_bytes / slice_bytes);
    int slice_shift = 0;
    while ((1ULL << slice_shift) < slice_bytes * 8) slice_shift++;
    const uint64_t cap = ((n_probes / n_slices + n_probes / n_slices / 8 + 65536) + 63) & ~63ULL;   // uniform random addresses: 12 % head room
    uint32_t* table = nullptr;
    uint2* queue = nullptr;
    uint64_t* answers = nullptr;
    unsigned long long* qcount = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    int rc = FGPU_OK;
    if (hipMalloc(&table, table_bytes) != hipSuccess || hipMalloc(&queue, (uint64_t)n_slices * cap * 8) != hipSuccess ||
        hipMalloc(&answers, ((uint64_t)n_slices * cap + n_probes) / 8 + 64) != hipSuccess || hipMalloc(&qcount, DIAG_MAX_SLICES * 8) != hipSuccess) {
        ctx->err = "diag: out of device memory";
        rc = FGPU_ERR_NOMEM;
    }
    if (!rc) {
        for (auto& e : ev) hipEventCreate(&e);
        hipMemsetAsync(table, 0x5A, table_bytes, ctx->stream);          // half of the bits set, like a filter at work
        const unsigned grid = FGPU_GRID_BLOCKS, bin_grid = (unsigned)(n_probes / (256 * DIAG_BIN_PER_THREAD));
        float direct = 0, bin = 0, probe = 0;
        for (int i = -1; i < iters; i++) {                               // one untimed round first
            const uint64_t salt = 0x5851F42D4C957F2DULL * (uint64_t)(i + 2);
            hipEventRecord(ev[0], ctx->stream);
            hipLaunchKernelGGL(k_diag_probe_direct, dim3(grid), dim3(256), 0, ctx->stream, (const uint32_t*)table, table_bytes * 8 - 1, n_probes, salt, answers);
            hipEventRecord(ev[1], ctx->stream);
            hipMemsetAsync(qcount, 0, DIAG_MAX_SLICES * 8, ctx->stream);
            hipLaunchKernelGGL(k_diag_bin, dim3(bin_grid), dim3(256), 0, ctx->stream, table_bytes * 8 - 1, slice_shift, n_slices, n_probes, salt, cap, qcount, queue);
            hipEventRecord(ev[2], ctx->stream);
            hipLaunchKernelGGL(k_diag_probe_binned, dim3(grid), dim3(256), 0, ctx->stream, (const uint32_t*)table, n_slices, cap, (const unsigned long long*)qcount,
                               (const uint2*)queue, answers + n_probes / 64);
            hipEventRecord(ev[3], ctx->stream);
            if (hipEventSynchronize(ev[3]) != hipSuccess) { rc = FGPU_ERR_HIP; ctx->err = "diag: binned probes failed"; break; }
            if (i >= 0) {
                float a = 0, b = 0, c = 0;
                hipEventElapsedTime(&a, ev[0], ev[1]);
                hipEventElapsedTime(&b, ev[1], ev[2]);
                hipEventElapsedTime(&c, ev[2], ev[3]);
                direct += a; bin += b; probe += c;
            }
        }
        if (!rc) {
            *direct_per_s = (double)n_probes * iters / (direct * 1e-3);
            *binned_per_s = (double)n_probes * iters / ((bin + probe) * 1e-3);
            *bin_ms = bin / iters;
            *probe_ms = probe / iters;
        }
        for (auto& e : ev) hipEventDestroy(e);
    }
    hipFree(table); hipFree(queue); hipFree(answers); hipFree(qcount);
    return rc;
}

// the chain A/B above: times per pass in ms (direct; bin, binned first level, rest of the survivors' chains), the survivors' share, and
// whether the two answer planes are equal bit for bit (`equal`).  fill_byte: every byte of the table (popcount / 8 = share of set bits).
extern "C" int fgpu_diag_binned_chain(fgpu_ctx* ctx, uint64_t table_bytes, uint64_t n_items, uint64_t slice_bytes, int n_hash, int fill_byte, int iters,
                                      double* direct_ms, double* bin_ms, double* first_ms, double* rest_ms, double* survivors_share, int* equal) {
    if (!ctx || !direct_ms || !bin_ms || !first_ms || !rest_ms || !survivors_share || !equal || iters < 1 || n_hash < 2 || n_hash > 8 ||
        table_bytes < (1u << 16) || (table_bytes & (table_bytes - 1)) || table_bytes > (1ULL << 29) || (slice_bytes & (slice_bytes - 1)) ||
        slice_bytes < 4096 || slice_bytes > table_bytes || table_bytes / slice_bytes > DIAG_MAX_SLICES || n_items < (1u << 16) || n_items > (1ULL << 31))
        return FGPU_ERR_ARG;
    n_items &= ~(uint64_t)(256 * DIAG_BIN_PER_THREAD - 1);
    const int n_slices = (int)(table_bytes / slice_bytes);
    int slice_shift = 0;
    while ((1ULL << slice_shift) < slice_bytes * 8) slice_shift++;
    const uint64_t cap = ((n_items / n_slices + n_items / n_slices / 8 + 65536) + 63) & ~63ULL;
    const uint64_t words = n_items / 64;
    uint32_t *table = nullptr, *survivors = nullptr;
    uint2* queue = nullptr;
    uint64_t *ans_a = nullptr, *ans_b = nullptr;
    unsigned long long *qcount = nullptr, *n_surv = nullptr;
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    int rc = FGPU_OK;
    if (hipMalloc(&table, table_bytes) != hipSuccess || hipMalloc(&queue, (uint64_t)n_slices * cap * 8) != hipSuccess ||
        hipMalloc(&survivors, n_items * 4) != hipSuccess || hipMalloc(&ans_a, words * 8) != hipSuccess || hipMalloc(&ans_b, words * 8) != hipSuccess ||
        hipMalloc(&qcount, DIAG_MAX_SLICES * 8) != hipSuccess || hipMalloc(&n_surv, 8) != hipSuccess) {
        ctx->err = "diag: out of device memory";
        rc = FGPU_ERR_NOMEM;
    }
    unsigned long long surv_host = 0;
    if (!rc) {
        for (auto& e : ev) hipEventCreate(&e);
        hipMemsetAsync(table, fill_byte & 0xFF, table_bytes, ctx->stream);
        const unsigned grid = FGPU_GRID_BLOCKS, bin_grid = (unsigned)(n_items / (256 * DIAG_BIN_PER_THREAD));
        const uint64_t bit_mask = table_bytes * 8 - 1;
        float t[4] = {0, 0, 0, 0};
        for (int i = -1; i < iters; i++) {
            const uint64_t salt = 0x5851F42D4C957F2DULL * (uint64_t)(i + 2);
            hipEventRecord(ev[0], ctx->stream);
            hipLaunchKernelGGL(k_diag_chain_direct, dim3(grid), dim3(256), 0, ctx->stream, (const uint32_t*)table, bit_mask, n_hash, n_items, salt, ans_a);
            hipEventRecord(ev[1], ctx->stream);
            hipMemsetAsync(qcount, 0, DIAG_MAX_SLICES * 8, ctx->stream);
            hipMemsetAsync(n_surv, 0, 8, ctx->stream);
            hipMemsetAsync(ans_b, 0, words * 8, ctx->stream);
            hipLaunchKernelGGL(k_diag_bin_items, dim3(bin_grid), dim3(256), 0, ctx->stream, bit_mask, slice_shift, n_slices, n_items, salt, cap, qcount, queue);
            hipEventRecord(ev[2], ctx->stream);
            hipLaunchKernelGGL(k_diag_first_binned, dim3(grid), dim3(256), 0, ctx->stream, (const uint32_t*)table, n_slices, cap,
                               (const unsigned long long*)qcount, (const uint2*)queue, survivors, n_surv);
            hipEventRecord(ev[3], ctx->stream);
            if (n_hash > 1)
                hipLaunchKernelGGL(k_diag_chain_rest, dim3(grid), dim3(256), 0, ctx->stream, (const uint32_t*)table, bit_mask, n_hash, salt,
                                   (const uint32_t*)survivors, (const unsigned long long*)n_surv, (unsigned long long*)ans_b);
            hipEventRecord(ev[4], ctx->stream);
            if (hipEventSynchronize(ev[4]) != hipSuccess) { rc = FGPU_ERR_HIP; ctx->err = "diag: binned chain failed"; break; }
            if (i >= 0) {
                float v;
                for (int k = 0; k < 4; k++) { hipEventElapsedTime(&v, ev[k], ev[k + 1]); t[k] += v; }
            }
        }
        if (!rc) {
            std::vector<uint64_t> a(words), b(words);
            hipMemcpy(a.data(), ans_a, words * 8, hipMemcpyDeviceToHost);
            hipMemcpy(b.data(), ans_b, words * 8, hipMemcpyDeviceToHost);
            hipMemcpy(&surv_host, n_surv, 8, hipMemcpyDeviceToHost);
            *equal = a == b ? 1 : 0;
            *direct_ms = t[0] / iters; *bin_ms = t[1] / iters; *first_ms = t[2] / iters; *rest_ms = t[3] / iters;
            *survivors_share = (double)surv_host / (double)n_items;
        }
        for (auto& e : ev) hipEventDestroy(e);
    }
    hipFree(table); hipFree(queue); hipFree(survivors); hipFree(ans_a); hipFree(ans_b); hipFree(qcount); hipFree(n_surv);
    return rc;
}

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

// the marking kernel's access pattern on given tables, nothing else of the kernel: per item three random 8-byte loads of filter words and -- six
// items in ten -- three atomicMin into the times of the same bit positions
template <int NH>
__global__ void __launch_bounds__(256) k_diag_mark_pattern(const uint2* __restrict__ pair, uint32_t* __restrict__ first, uint64_t bit_mask, uint64_t n, uint64_t salt,
                                                           uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = diag_rand(i ^ salt);
        const uint64_t hA = r & bit_mask, hB = diag_rand(r) | 1;
        uint64_t h = hA;
        uint2 v[NH];
#pragma unroll
        for (int q = 0; q < NH; q++) { v[q] = pair[h >> 5]; h = (h + hB) & bit_mask; }
#pragma unroll
        for (int q = 0; q < NH; q++) acc += v[q].x ^ v[q].y;
        if (diag_rand(r + 7) % 10 < 6) {
            h = hA;
#pragma unroll
            for (int q = 0; q < NH; q++) { atomicMin(&first[h], 0xFFFFFFF0u + (uint32_t)(i & 7)); h = (h + hB) & bit_mask; }
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

static void launch_mark_pattern(int nh, hipStream_t st, const void* pair, uint32_t* first, uint64_t bit_mask, uint64_t n, uint64_t salt, uint32_t* sink) {
    const dim3 g(fgpu_grid(n, 256)), b(256);
    if (nh <= 2) hipLaunchKernelGGL(k_diag_mark_pattern<2>, g, b, 0, st, (const uint2*)pair, first, bit_mask, n, salt, sink);
    else if (nh == 3) hipLaunchKernelGGL(k_diag_mark_pattern<3>, g, b, 0, st, (const uint2*)pair, first, bit_mask, n, salt, sink);
    else hipLaunchKernelGGL(k_diag_mark_pattern<4>, g, b, 0, st, (const uint2*)pair, first, bit_mask, n, salt, sink);
}

// The same two access patterns on the context's OWN tables of a load pass (scripts/kinds_probe.py, VERDICT r4 item 4: the marking kernel on 2^33-bit
// filters runs at one of a few speeds that stay with an allocation): random 4-byte loads from the interleaved filter pair, random atomicMin into the
// first-set times.  Between passes only: both tables are rewritten by the next fgpu_load_begin.
extern "C" int fgpu_diag_load_tables(fgpu_ctx* ctx, uint64_t n_access, double* pair_loads_per_s, double* first_atomics_per_s, double* mixed_items_per_s) {
    if (!ctx || !pair_loads_per_s || !first_atomics_per_s || n_access < 8) return FGPU_ERR_ARG;
    if (ctx->phase != 0 || !ctx->pair || !ctx->first) { ctx->err = "fgpu_diag_load_tables: between passes, after a load pass in the pair layout"; return FGPU_ERR_STATE; }
    FGPU_HIP(hipSetDevice(ctx->prm.device));
    uint32_t* sink = nullptr;
    hipEvent_t e[4];
    FGPU_HIP(hipMalloc(&sink, 64));
    for (hipEvent_t& x : e) hipEventCreate(&x);
    const uint64_t rounds = n_access / 8;
    const unsigned grid = fgpu_grid(rounds, 256);
    for (int i = -1; i < 1; i++) {          // (one warm-up, one timed)
        if (i == 0) hipEventRecord(e[0], ctx->stream);
        hipLaunchKernelGGL(k_diag_random<0>, dim3(grid), dim3(256), 0, ctx->stream, (uint32_t*)ctx->pair, ctx->bloom_bytes * 2 / 4 - 1, rounds, 77ULL + (uint64_t)i, sink);
    }
    hipEventRecord(e[1], ctx->stream);
    hipLaunchKernelGGL(k_diag_random<1>, dim3(grid), dim3(256), 0, ctx->stream, ctx->first, ctx->prm.tai - 1, rounds, 99ULL, sink);
    hipLaunchKernelGGL(k_diag_random<1>, dim3(grid), dim3(256), 0, ctx->stream, ctx->first, ctx->prm.tai - 1, rounds, 101ULL, sink);
    hipEventRecord(e[2], ctx->stream);
    const uint64_t items = n_access / 4;
    launch_mark_pattern(ctx->fd.n_hash, ctx->stream, ctx->pair, ctx->first, ctx->prm.tai - 1, items, 5ULL, sink);
    hipEventRecord(e[3], ctx->stream);
    const hipError_t err = hipEventSynchronize(e[3]);
    float ms_l = 0, ms_a = 0, ms_m = 0;
    hipEventElapsedTime(&ms_l, e[0], e[1]);
    hipEventElapsedTime(&ms_a, e[1], e[2]);
    hipEventElapsedTime(&ms_m, e[2], e[3]);
    if (mixed_items_per_s) *mixed_items_per_s = (double)items / (ms_m * 1e-3);
    for (hipEvent_t& x : e) hipEventDestroy(x);
    hipFree(sink);
    FGPU_HIP(err);
    *pair_loads_per_s = (double)(rounds * 8) / (ms_l * 1e-3);
    *first_atomics_per_s = (double)(rounds * 16) / (ms_a * 1e-3);
    return FGPU_OK;
}

// ---- where the filter pair lies (round 5) ---------------------------------------------------------------------------------------
// On 2^33-bit filters the marking kernel ran at one of two speeds, 236 or 275 ms per 25 M reads of config 4, fixed for the life of a context and
// changing from context to context.  Counters of a fast and a slow context agree in every COUNT (read requests, atomics, DRAM reads and writes, TLB
// requests and misses, L2 hits) and differ in how long a request stays outstanding at the memory side (TCC_EA0_RDREQ_LEVEL / _RDREQ 2 475 -> 2 810
// cycles, _ATOMIC_LEVEL / _ATOMIC 1 257 -> 1 404: profiles/r05_load_mark_kinds.txt).  Random loads from the pair alone and random atomicMin into
// first[] alone run at the same rates in both; the two TOGETHER, as the kernel mixes them, do not (6.9e9 against 6.3e9 items/s) -- and the same
// first[] with ANOTHER allocation of the pair is fast again: what decides is where the 2 GiB of filter words lie relative to the 32 GiB of times.
// The driver's placement cannot be asked for, so it is measured: after the two allocations of a pass's first fgpu_load_begin the mixed pattern is
// timed on them (2^24 items, ~2.5 ms); if the two patterns do not overlap (rate below 1.04 x the rate their separate rates add up to), further
// allocations of the pair are tried -- held meanwhile, so that they differ -- until one does (16 at most), and the rest are returned.
// Filters of 2^32 bits and more only (below, the pair lives in the Infinity Cache).  FGPU_PAIR_PLACE=0 switches it off, =1 on for every size (tests),
// FGPU_DEBUG_PLACE=1 tells what it measured.
int fgpu_place_pair(fgpu_ctx* ctx) {
    static const char* env = getenv("FGPU_PAIR_PLACE");
    static const bool tell = getenv("FGPU_DEBUG_PLACE") != nullptr;
    if (env ? env[0] == '0' : ctx->prm.tai < (1ULL << 32)) return FGPU_OK;
    if (!ctx->pair || !ctx->first) return FGPU_OK;
    const int nh = std::min(std::max(ctx->fd.n_hash, 2), 4);
    const uint64_t n_items = 1ULL << 24, mask = ctx->prm.tai - 1, pair_bytes = ctx->bloom_bytes * 2;
    uint32_t* sink = nullptr;
    FGPU_HIP(hipMalloc(&sink, 64));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto timed = [&](const std::function<void(uint64_t)>& launch, double* ms_out) -> int {
        launch(1);                                     // (warm-up: the pages' translations)
        hipEventRecord(e0, ctx->stream);
        launch(2);
        hipEventRecord(e1, ctx->stream);
        if (hipEventSynchronize(e1) != hipSuccess) { ctx->err = "placing the filter pair: a probe failed"; return FGPU_ERR_HIP; }
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        *ms_out = ms;
        return FGPU_OK;
    };
    auto mixed = [&](void* pair, double* rate) -> int {
        double ms = 0;
        const int rc = timed([&](uint64_t salt) { launch_mark_pattern(nh, ctx->stream, pair, ctx->first, mask, n_items, salt, sink); }, &ms);
        *rate = (double)n_items / (ms * 1e-3);
        return rc;
    };
    int rc = FGPU_OK;
    double ms_l = 0, ms_a = 0, r0 = 0;
    const uint64_t rounds = (1ULL << 25) / 8;
    rc = timed([&](uint64_t salt) { hipLaunchKernelGGL(k_diag_random<0>, dim3(fgpu_grid(rounds, 256)), dim3(256), 0, ctx->stream, (uint32_t*)ctx->pair, pair_bytes / 4 - 1, rounds, salt, sink); }, &ms_l);
    if (!rc) rc = timed([&](uint64_t salt) { hipLaunchKernelGGL(k_diag_random<1>, dim3(fgpu_grid(rounds, 256)), dim3(256), 0, ctx->stream, ctx->first, mask, rounds, salt, sink); }, &ms_a);
    if (!rc) rc = mixed(ctx->pair, &r0);
    std::vector<void*> held;
    if (!rc) {
        const double loads = (double)(rounds * 8) / (ms_l * 1e-3), atomics = (double)(rounds * 8) / (ms_a * 1e-3);
        const double serial = 1.0 / ((double)nh / loads + 0.6 * (double)nh / atomics);     // items/s if the two patterns did not overlap at all
        const double want = 1.04 * serial;
        void* best = ctx->pair;
        double best_rate = r0;
        int tried = 0;
        while (best_rate < want && tried < 16) {
            void* cand = nullptr;
            if (hipMalloc(&cand, pair_bytes) != hipSuccess) { (void)hipGetLastError(); break; }
            held.push_back(cand);
            tried++;
            double r = 0;
            if ((rc = mixed(cand, &r))) break;
            if (r > best_rate) { best = cand; best_rate = r; }
        }
        if (tell)
            fprintf(stderr, "[place] filter pair: loads %.3g/s, atomics %.3g/s, together %.3g items/s as allocated (no overlap: %.3g); %d other allocations tried, kept %s at %.3g\n",
                    loads, atomics, r0, serial, tried, best == (void*)ctx->pair ? "the first" : "another", best_rate);
        if (best != (void*)ctx->pair) {
            held.push_back(ctx->pair);
            ctx->pair = (uint2*)best;
        }
        for (void* p : held) if (p != (void*)ctx->pair) hipFree(p);
        held.clear();
    }
    for (void* p : held) if (p != (void*)ctx->pair) hipFree(p);
    hipFree(sink);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return rc;
}

// Does the speed of the mixed pattern depend on where the PAIR array lies (2 x the filter: cheap to allocate again) while the first-set times stay where
// they are?  k candidate allocations held at once (so that they differ), each probed with the context's first[]; items per second each.
extern "C" int fgpu_diag_pair_placements(fgpu_ctx* ctx, int k, uint64_t n_items, double* items_per_s) {
    if (!ctx || k < 1 || k > 16 || !items_per_s || n_items < 64) return FGPU_ERR_ARG;
    if (ctx->phase != 0 || !ctx->first) { ctx->err = "fgpu_diag_pair_placements: between passes, after a load pass in the pair layout"; return FGPU_ERR_STATE; }
    FGPU_HIP(hipSetDevice(ctx->prm.device));
    std::vector<void*> cand((size_t)k, nullptr), spacers;
    uint32_t* sink = nullptr;
    FGPU_HIP(hipMalloc(&sink, 64));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    int rc = FGPU_OK;
    for (int i = 0; i < k && rc == FGPU_OK; i++) {
        // (from the eighth candidate on a spacer of growing size goes in front: 64 MiB, 128 MiB ... -- another offset, not just the next block)
        if (i >= 8) {
            void* spacer = nullptr;
            if (hipMalloc(&spacer, (64ULL << 20) << (i - 8)) == hipSuccess) spacers.push_back(spacer); else (void)hipGetLastError();
        }
        if (hipMalloc(&cand[(size_t)i], ctx->bloom_bytes * 2) != hipSuccess) { (void)hipGetLastError(); cand[(size_t)i] = nullptr; ctx->err = "diag: out of device memory"; rc = FGPU_ERR_NOMEM; break; }
        hipMemsetAsync(cand[(size_t)i], 0, ctx->bloom_bytes * 2, ctx->stream);
        for (int rep = 0; rep < 2; rep++) {
            if (rep == 1) hipEventRecord(e0, ctx->stream);
            launch_mark_pattern(ctx->fd.n_hash, ctx->stream, cand[(size_t)i], ctx->first, ctx->prm.tai - 1, n_items, 11ULL + (uint64_t)rep, sink);
        }
        hipEventRecord(e1, ctx->stream);
        if (hipEventSynchronize(e1) != hipSuccess) { ctx->err = "diag: probe failed"; rc = FGPU_ERR_HIP; break; }
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        items_per_s[i] = (double)n_items / (ms * 1e-3);
    }
    for (void* p : cand) if (p) hipFree(p);
    for (void* p : spacers) hipFree(p);
    hipFree(sink);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return rc;
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
