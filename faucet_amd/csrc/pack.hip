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

// one thread per 64 stream positions: 1 bad word + 2 code words, plain stores
__global__ void __launch_bounds__(256) k_pack(const unsigned char* __restrict__ bases, const uint64_t* __restrict__ offs,
                                              uint64_t n_reads, uint64_t T, uint64_t n_words, uint64_t* __restrict__ codes,
                                              uint64_t* __restrict__ bad, unsigned char* __restrict__ readflag) {
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_words + FGPU_PADW) return;
    if (g >= n_words) {   // padding words so that funnel reads past the end see "bad"
        bad[g] = ~0ULL;
        codes[2 * g] = 0;
        codes[2 * g + 1] = 0;
        return;
    }
    const uint64_t off0 = offs[0];
    uint64_t s0 = g * 64;
    // largest i with S_i <= s0
    uint64_t lo = 0, hi = n_reads - 1;
    while (lo < hi) {
        uint64_t mid = (lo + hi + 1) >> 1;
        uint64_t S = (offs[mid] - off0) + mid;
        if (S <= s0) lo = mid; else hi = mid - 1;
    }
    uint64_t i = lo;
    uint64_t rbeg = offs[i], rend = offs[i + 1];
    uint64_t S = (rbeg - off0) + i;
    uint64_t c = s0 - S;              // character index inside read i (== len: the separator)
    uint64_t len = rend - rbeg;
    uint64_t badw = 0, cw0 = 0, cw1 = 0;
    for (int q = 0; q < 64; q++) {
        uint64_t s = s0 + q;
        int code = 0;
        bool isbad = true;
        if (s < T) {
            if (c < len) {
                unsigned char ch = bases[rbeg + c];
                if (is_acgt(ch)) { isbad = false; code = (ch >> 1) & 3; }
                else readflag[i] = 1;   // benign same-value race between the threads sharing read i
                c++;
            } else {   // separator after read i; move on to read i+1
                i++;
                if (i < n_reads) { rbeg = rend; rend = offs[i + 1]; len = rend - rbeg; }
                c = 0;
            }
        }
        if (isbad) badw |= 1ULL << q;
        if (q < 32) cw0 |= (uint64_t)code << (62 - 2 * q);
        else cw1 |= (uint64_t)code << (62 - 2 * (q - 32));
    }
    bad[g] = badw;
    codes[2 * g] = cw0;
    codes[2 * g + 1] = cw1;
}

// one thread per read that has interior bad characters: rewrite its positions in reverse token order
__global__ void __launch_bounds__(256) k_pack_fix(const unsigned char* __restrict__ bases, const uint64_t* __restrict__ offs,
                                                  uint64_t n_reads, unsigned long long* codes, unsigned long long* bad,
                                                  const unsigned char* __restrict__ readflag, unsigned long long* max_len) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    {   // longest read of the batch (one atomic per wave)
        unsigned long long l = i < n_reads ? (unsigned long long)(offs[i + 1] - offs[i]) : 0;
        for (int o = 32; o > 0; o >>= 1) { unsigned long long t = __shfl_down(l, o, 64); l = t > l ? t : l; }
        if ((threadIdx.x & 63) == 0 && l) atomicMax(max_len, l);
    }
    if (i >= n_reads || !readflag[i]) return;
    const uint64_t off0 = offs[0];
    uint64_t rbeg = offs[i], L = offs[i + 1] - rbeg;
    uint64_t S = (rbeg - off0) + i;
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
    uint64_t total;
    if (reads->on_device) {
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
    FGPU_LAUNCH("pack", k_pack, fgpu_blocks(bb.n_words + FGPU_PADW, 256), 256, d_bases, d_offs, n, T, bb.n_words, (uint64_t*)bb.codes.p,
                (uint64_t*)bb.bad.p, (unsigned char*)bb.readflag.p);
    FGPU_LAUNCH("pack_fix", k_pack_fix, fgpu_blocks(n, 256), 256, d_bases, d_offs, n, (unsigned long long*)bb.codes.p,
                (unsigned long long*)bb.bad.p, (const unsigned char*)bb.readflag.p, &ctx->counters->max_read_len);
    return FGPU_OK;
}
