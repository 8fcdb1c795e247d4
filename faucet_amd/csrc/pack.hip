// pack.hip — turn a batch of reads into the normalized 2-bit stream both passes work on.
//
// Replaces the per-read string handling of the reference:
//   getUnambiguousReads (utils/Kmer.cpp:64-80)   split at every byte that is not A C G T, keep pieces,
//                                                 hand them out LAST FIRST (push_front, :77)
//   NT2int              (utils/Kmer.cpp:82-88)   (c >> 1) & 3
//
// Stream layout: read i occupies positions S_i .. S_i+len_i-1 followed by ONE separator position,
// S_i = (offsets[i] - offsets[0]) + i.  A separator and every non-ACGT byte are "bad" positions.
// Inside a read that has bad characters the tokens (maximal runs of good / bad characters) are laid
// out in REVERSE token order, so that ascending stream position == the reference's processing order
// (reads in file order, segments of a read last first, windows of a segment ascending).  Every later
// kernel therefore needs no read or segment table: a k-mer window is valid iff its k positions hold no
// bad bit, and its processing time is its stream position.
#include "fgpu_ctx.h"

namespace {

__device__ __forceinline__ bool is_acgt(unsigned char c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }

// 32 mask bits -> the even bits of a 64-bit word (bit j -> bit 2j)
__device__ __forceinline__ uint64_t spread32(uint32_t v) {
    uint64_t x = v;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFULL;
    x = (x | (x << 8)) & 0x00FF00FF00FF00FFULL;
    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0FULL;
    x = (x | (x << 2)) & 0x3333333333333333ULL;
    x = (x | (x << 1)) & 0x5555555555555555ULL;
    return x;
}

// One LANE per stream position, one wave per 64-position word: the bases are read as coalesced 64-byte rows, the bad
// word and the two code-bit planes come out of three ballots.  Every wave takes a contiguous run of words; the read
// that contains a position is found once per run (binary search) and then followed with a cursor over a window of 64
// read start positions held one per lane, so that the only memory access per word is the row of bases itself.
__global__ void __launch_bounds__(256) k_pack(const unsigned char* __restrict__ bases, const uint64_t* __restrict__ offs,
                                              const uint64_t* __restrict__ starts, uint64_t n_reads, uint64_t T, uint64_t n_words, uint64_t* __restrict__ codes,
                                              uint64_t* __restrict__ bad, unsigned char* __restrict__ readflag) {
    const uint64_t total_words = n_words + FGPU_PADW;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t per = (total_words + n_waves - 1) / n_waves;
    const uint64_t w0 = wave * per, w1 = w0 + per < total_words ? w0 + per : total_words;
    if (w0 >= w1) return;
    const int lane = fd_lane();
    const uint64_t off0 = offs[0];
    const uint64_t NEVER = ~0ULL;
    // largest i with S_i <= first position of the run (S_i = offs[i] - off0 + i; S_n = T); the same in every lane
    uint64_t cur = 0;
    {
        const uint64_t s0 = w0 * 64;
        uint64_t lo = 0, hi = n_reads - 1;
        while (lo < hi) {
            uint64_t mid = (lo + hi + 1) >> 1;
            if ((offs[mid] - off0) + mid <= s0) lo = mid; else hi = mid - 1;
        }
        cur = lo;
    }
    uint64_t base = NEVER, Sreg = 0;   // lane l holds S_{base+l}
    constexpr int U = 4;               // words per trip: the U rows of bases are loaded together (independent loads in flight)
    for (uint64_t wb = w0; wb < w1; wb += U) {
        uint64_t addr[U], rd[U];
        bool chr[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t w = wb + u;
            chr[u] = false;
            addr[u] = 0;
            rd[u] = 0;
            if (w >= w1 || w >= n_words) continue;
            if (base == NEVER || cur + 2 > base + 63) {
                base = cur;
                const uint64_t j = base + lane;
                Sreg = j <= n_reads ? (offs[j] - off0) + j : NEVER;
            }
            const int r = (int)(cur - base);
            const uint64_t S0 = __shfl(Sreg, r, 64), S1 = __shfl(Sreg, r + 1, 64), S2 = __shfl(Sreg, r + 2, 64);
            const uint64_t s = w * 64 + lane;
            uint64_t i, Si, Sn;   // read containing position s (its characters or its separator), its start, the next start
            if (s < S1) { i = cur; Si = S0; Sn = S1; }
            else if (s < S2) { i = cur + 1; Si = S1; Sn = S2; }
            else {   // reads shorter than a word: follow the offsets in memory
                i = cur + 2; Si = S2;
                for (;;) {
                    Sn = i + 1 <= n_reads ? (offs[i + 1] - off0) + (i + 1) : NEVER;
                    if (s < Sn) break;
                    i++;
                    Si = Sn;
                }
            }
            chr[u] = s < T && s + 1 < Sn;   // a character of read i (s + 1 == Sn: its separator)
            // first byte of read i: offs[i] for contiguous batches, starts[i] when the reads lie inside raw text
            addr[u] = (starts ? (chr[u] ? starts[i] : 0) : Si + off0 - i) + (s - Si);
            rd[u] = i;
            cur = __shfl(i, 63, 64);
            if (cur > n_reads - 1) cur = n_reads - 1;
        }
        unsigned char ch[U];
#pragma unroll
        for (int u = 0; u < U; u++) ch[u] = chr[u] ? bases[addr[u]] : (unsigned char)0;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t w = wb + u;
            if (w >= w1) continue;
            if (w >= n_words) {   // padding words so that funnel reads past the end see "bad"
                if (lane == 0) { bad[w] = ~0ULL; codes[2 * w] = 0; codes[2 * w + 1] = 0; }
                continue;
            }
            int code = 0;
            bool isbad = true;
            if (chr[u]) {
                if (is_acgt(ch[u])) { isbad = false; code = (ch[u] >> 1) & 3; }
                else readflag[rd[u]] = 1;   // benign same-value race between the lanes sharing a read
            }
            const uint64_t badw = __ballot(isbad), hi = __ballot(code & 2), lo = __ballot(code & 1);
            if (lane < 2) {   // lane t assembles code word t: position q of its half -> bits 63-2q (high code bit), 62-2q (low)
                const uint32_t h32 = (uint32_t)(hi >> (32 * lane)), l32 = (uint32_t)(lo >> (32 * lane));
                codes[2 * w + lane] = (spread32(__brev(h32)) << 1) | spread32(__brev(l32));
                if (lane == 0) bad[w] = badw;
            }
        }
    }
}

// one thread per read that has interior bad characters: rewrite its positions in reverse token order
__global__ void __launch_bounds__(256) k_pack_fix(const unsigned char* __restrict__ bases, const uint64_t* __restrict__ offs,
                                                  const uint64_t* __restrict__ starts, uint64_t n_reads, unsigned long long* codes, unsigned long long* bad,
                                                  const unsigned char* __restrict__ readflag, unsigned long long* max_len) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    {   // longest read of the batch: an atomic only when the wave raises the maximum (same-address atomics serialise)
        unsigned long long l = i < n_reads ? (unsigned long long)(offs[i + 1] - offs[i]) : 0;
        for (int o = 32; o > 0; o >>= 1) { unsigned long long t = __shfl_down(l, o, 64); l = t > l ? t : l; }
        if ((threadIdx.x & 63) == 0 && l > *(volatile unsigned long long*)max_len) atomicMax(max_len, l);
    }
    if (i >= n_reads || !readflag[i]) return;
    const uint64_t off0 = offs[0];
    const uint64_t L = offs[i + 1] - offs[i];
    const uint64_t S = (offs[i] - off0) + i;
    const uint64_t rbeg = starts ? starts[i] : offs[i];
    uint64_t s = 0;
    while (s < L) {
        bool good = is_acgt(bases[rbeg + s]);
        uint64_t e = s + 1;
        while (e < L && is_acgt(bases[rbeg + e]) == good) e++;
        // token [s, e) moves to [L - e, L - s)
        for (uint64_t c = s; c < e; c++) {
            uint64_t p = S + (c + L - e - s);
            unsigned long long bbit = 1ULL << (p & 63);
            int sh = 62 - 2 * (int)(p & 31);
            unsigned long long cmask = 3ULL << sh;
            if (good) {
                unsigned long long code = (unsigned long long)((bases[rbeg + c] >> 1) & 3);
                atomicAnd(&bad[p >> 6], ~bbit);
                atomicAnd(&codes[p >> 5], ~cmask);
                atomicOr(&codes[p >> 5], code << sh);
            } else {
                atomicOr(&bad[p >> 6], bbit);
                atomicAnd(&codes[p >> 5], ~cmask);
            }
        }
        s = e;
    }
}

}  // namespace

int fgpu_stage_pack(fgpu_ctx* ctx, const fgpu_reads* reads) {
    BatchBufs& bb = *ctx->cur;
    const uint64_t n = reads->n_reads;
    const unsigned char* d_bases;
    const uint64_t* d_offs;
    const uint64_t* d_starts = nullptr;
    uint64_t total;
    if (reads->starts && !reads->on_device) { ctx->err = "fgpu_reads.starts needs a device batch"; return FGPU_ERR_ARG; }
    if (reads->on_device) {
        d_starts = reads->starts;
        uint64_t ends[2];
        FGPU_HIP(hipMemcpyAsync(&ends[0], reads->offsets, 8, hipMemcpyDeviceToHost, ctx->stream));
        FGPU_HIP(hipMemcpyAsync(&ends[1], reads->offsets + n, 8, hipMemcpyDeviceToHost, ctx->stream));
        FGPU_HIP(hipStreamSynchronize(ctx->stream));
        total = ends[1] - ends[0];
        d_bases = (const unsigned char*)reads->bases;
        d_offs = reads->offsets;
    } else {
        total = reads->offsets[n] - reads->offsets[0];
        int rc = fgpu_ensure(ctx, &bb.in_bases, total + 16);
        if (rc) return rc;
        rc = fgpu_ensure(ctx, &bb.in_offsets, (n + 1) * 8);
        if (rc) return rc;
        // bases are copied from offsets[0] on, so the device copy is addressed with the same offsets
        // shifted by offsets[0]: keep the original offsets and bias the base pointer instead.
        FGPU_HIP(hipMemcpyAsync(bb.in_bases.p, reads->bases + reads->offsets[0], total, hipMemcpyHostToDevice, ctx->stream));
        FGPU_HIP(hipMemcpyAsync(bb.in_offsets.p, reads->offsets, (n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        d_bases = (const unsigned char*)bb.in_bases.p - reads->offsets[0];
        d_offs = (const uint64_t*)bb.in_offsets.p;
    }
    const uint64_t T = total + n;
    uint64_t maxb = ctx->prm.max_batch_bases;
    if (T > maxb || T >= 0xFFFFFF00ULL) {
        ctx->err = "batch exceeds max_batch_bases";
        return FGPU_ERR_CAPACITY;
    }
    bb.T = T;
    bb.n_words = (T + 63) / 64;
    bb.n_reads = n;
    bb.d_offs = d_offs;
    int rc;
    if (n == 0) {   // nothing to pack; later stages see an empty stream
        if ((rc = fgpu_ensure(ctx, &bb.codes, 64))) return rc;
        if ((rc = fgpu_ensure(ctx, &bb.bad, 64))) return rc;
        FGPU_HIP(hipMemsetAsync(bb.bad.p, 0xFF, 64, ctx->stream));
        FGPU_HIP(hipMemsetAsync(bb.codes.p, 0, 64, ctx->stream));
        return FGPU_OK;
    }
    if ((rc = fgpu_ensure(ctx, &bb.codes, (2 * (bb.n_words + FGPU_PADW)) * 8))) return rc;
    if ((rc = fgpu_ensure(ctx, &bb.bad, (bb.n_words + FGPU_PADW) * 8))) return rc;
    if ((rc = fgpu_ensure(ctx, &bb.readflag, n + 16))) return rc;
    FGPU_HIP(hipMemsetAsync(bb.readflag.p, 0, n, ctx->stream));
    FGPU_LAUNCH("pack", k_pack, fgpu_grid((bb.n_words + FGPU_PADW) * 8, 256), 256, d_bases, d_offs, d_starts, n, T, bb.n_words, (uint64_t*)bb.codes.p,
                (uint64_t*)bb.bad.p, (unsigned char*)bb.readflag.p);
    FGPU_LAUNCH("pack_fix", k_pack_fix, fgpu_blocks(n, 256), 256, d_bases, d_offs, d_starts, n, (unsigned long long*)bb.codes.p,
                (unsigned long long*)bb.bad.p, (const unsigned char*)bb.readflag.p, &ctx->counters->max_read_len);
    return FGPU_OK;
}
