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

// bit 7 of every byte of v that is zero (exact)
__device__ __forceinline__ uint64_t zero_bytes(uint64_t v) {
    return ~(((v & 0x7F7F7F7F7F7F7F7FULL) + 0x7F7F7F7F7F7F7F7FULL) | v) & 0x8080808080808080ULL;
}

// 8 characters (first in the low byte) -> their 2-bit codes (character q at bits 2q..2q+1) and "is A/C/G/T" bits (bit q)
__device__ __forceinline__ void convert8(uint64_t x, uint32_t& code16, uint32_t& good8) {
    uint64_t t = (x >> 1) & 0x0303030303030303ULL;
    t = (t | (t >> 6)) & 0x000F000F000F000FULL;
    t = (t | (t >> 12)) & 0x000000FF000000FFULL;
    t = (t | (t >> 24)) & 0xFFFFULL;
    code16 = (uint32_t)t;
    const uint64_t g = zero_bytes(x ^ 0x4141414141414141ULL) | zero_bytes(x ^ 0x4343434343434343ULL) |
                       zero_bytes(x ^ 0x4747474747474747ULL) | zero_bytes(x ^ 0x5454545454545454ULL);
    good8 = (uint32_t)(((g >> 7) * 0x0102040810204080ULL) >> 56);
}

// One LANE per 32 stream positions = one 64-bit word of codes and half a word of the bad plane.  The positions of a lane are
// at most a few runs of consecutive characters (a read, its separator, the next read ...); a run is fetched as the 16-byte
// aligned chunks that hold it, funnel-shifted to its first byte and converted 8 characters at a time with SWAR arithmetic
// (about a quarter of an instruction per position; one lane per position with ballots cost ten times that and ran at 5 % of
// the streaming rate).  Every wave takes a contiguous run of lanes' worth of positions; the read that contains a position is
// found by bisection over a window of 64 read start positions held one per lane (registers, no memory access), or over the
// offsets in memory where reads are so short that 64 of them do not span the 2048 positions of a trip.
__global__ void __launch_bounds__(256) k_pack(const unsigned char* __restrict__ bases, const uint64_t* __restrict__ offs,
                                              const uint64_t* __restrict__ starts, uint64_t n_reads, uint64_t T, uint64_t n_words, uint64_t* __restrict__ codes,
                                              uint32_t* __restrict__ bad32, unsigned char* __restrict__ readflag) {
    const uint64_t total_hw = 2 * (n_words + FGPU_PADW);
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t per = (((total_hw + n_waves - 1) / n_waves) + 63) & ~63ULL;
    const uint64_t h0 = wave * per, h1 = h0 + per < total_hw ? h0 + per : total_hw;
    if (h0 >= h1) return;
    const int lane = fd_lane();
    const uint64_t off0 = offs[0];
    const uint64_t NEVER = ~0ULL;
    // largest i with S_i <= first position of the run (S_i = offs[i] - off0 + i; S_n = T); the same in every lane
    uint64_t cur = 0;
    {
        const uint64_t s_first = h0 * 32;
        uint64_t lo = 0, hi = n_reads - 1;
        while (lo < hi) {
            uint64_t mid = (lo + hi + 1) >> 1;
            if ((offs[mid] - off0) + mid <= s_first) lo = mid; else hi = mid - 1;
        }
        cur = lo;
    }
    uint64_t base = NEVER, Sreg = 0;   // lane l holds S_{base+l}
    for (uint64_t hb = h0; hb < h1; hb += 64) {
        const uint64_t hw = hb + lane;
        const uint64_t s0 = hw * 32;
        const uint64_t last_s0 = ((hb + 63 < h1 ? hb + 63 : h1 - 1)) * 32;
        if (base == NEVER || __shfl(Sreg, 63, 64) <= last_s0) {   // the window does not reach the end of this trip: start it at cur
            base = cur;
            const uint64_t j = base + lane;
            Sreg = j <= n_reads ? (offs[j] - off0) + j : NEVER;
        }
        uint64_t i;
        if (__shfl(Sreg, 63, 64) > last_s0) {
            int r = 0;
#pragma unroll
            for (int step = 32; step > 0; step >>= 1) {
                const uint64_t v = __shfl(Sreg, r + step, 64);
                if (v <= s0) r += step;
            }
            i = base + (uint64_t)r;
        } else {                 // reads shorter than 32: bisection over the offsets themselves
            uint64_t lo = cur, hi = n_reads - 1;
            while (lo < hi) {
                uint64_t mid = (lo + hi + 1) >> 1;
                if ((offs[mid] - off0) + mid <= s0) lo = mid; else hi = mid - 1;
            }
            i = lo;
        }
        if (s0 >= T || i > n_reads - 1) i = n_reads - 1;
        cur = __shfl(i, 63, 64);

        uint64_t out_code = 0;
        uint32_t out_good = 0;
        if (hw < h1 && s0 < T) {
            const uint64_t end = s0 + 32 < T ? s0 + 32 : T;
            uint64_t pos = s0;
            uint64_t Si = (offs[i] - off0) + i;
            while (pos < end) {
                const uint64_t Sn = (offs[i + 1] - off0) + (i + 1);   // i <= n_reads - 1; S_n = T
                const uint64_t char_end = Sn - 1;                     // the separator behind read i
                if (pos < char_end) {
                    const uint64_t pos_b = char_end < end ? char_end : end;
                    const uint32_t n = (uint32_t)(pos_b - pos), d = (uint32_t)(pos - s0);
                    // first byte of read i: offs[i] for contiguous batches, starts[i] when the reads lie inside raw text
                    const uint64_t A = (uint64_t)(uintptr_t)bases + (starts ? starts[i] : Si + off0 - i) + (pos - Si);
                    const ulonglong2* c0 = (const ulonglong2*)(A & ~15ULL);
                    const uint32_t sh = (uint32_t)(A & 15), need_bytes = sh + n;   // chunks that hold a needed byte are safe to load
                    ulonglong2 q0 = c0[0], q1 = make_ulonglong2(0, 0), q2 = make_ulonglong2(0, 0);
                    if (need_bytes > 16) q1 = c0[1];
                    if (need_bytes > 32) q2 = c0[2];
                    uint64_t b0 = q0.x, b1 = q0.y, b2 = q1.x, b3 = q1.y, b4 = q2.x, b5 = q2.y;
                    if (sh & 8) { b0 = b1; b1 = b2; b2 = b3; b3 = b4; b4 = b5; }
                    const uint32_t bs = (sh & 7) * 8;
                    if (bs) {
                        b0 = (b0 >> bs) | (b1 << (64 - bs));
                        b1 = (b1 >> bs) | (b2 << (64 - bs));
                        b2 = (b2 >> bs) | (b3 << (64 - bs));
                        b3 = (b3 >> bs) | (b4 << (64 - bs));
                    }
                    uint32_t c16[4], g8[4];
                    convert8(b0, c16[0], g8[0]);
                    convert8(b1, c16[1], g8[1]);
                    convert8(b2, c16[2], g8[2]);
                    convert8(b3, c16[3], g8[3]);
                    const uint32_t keep = n == 32 ? ~0u : (1u << n) - 1;
                    const uint32_t good = (g8[0] | (g8[1] << 8) | (g8[2] << 16) | (g8[3] << 24)) & keep;
                    uint64_t cw = (uint64_t)c16[0] | ((uint64_t)c16[1] << 16) | ((uint64_t)c16[2] << 32) | ((uint64_t)c16[3] << 48);
                    const uint64_t gs = spread32(good);
                    cw &= gs | (gs << 1);                                  // characters that are not A/C/G/T carry code 0
                    // character q at bits 2q..2q+1 -> first character in the two most significant bits
                    uint64_t y = __builtin_bitreverse64(cw);
                    y = ((y >> 1) & 0x5555555555555555ULL) | ((y & 0x5555555555555555ULL) << 1);
                    out_code |= y >> (2 * d);
                    out_good |= good << d;
                    if (good != keep) readflag[i] = 1;                     // benign same-value race between the lanes sharing a read
                    pos = pos_b;
                }
                if (pos == char_end) pos++;
                if (pos >= Sn) { i++; Si = Sn; }
            }
        }
        if (hw < h1) {
            codes[hw] = out_code;
            bad32[hw] = ~out_good;
        }
    }
}

// one thread per read that has interior bad characters: rewrite its positions in reverse token order
__global__ void __launch_bounds__(256) k_pack_fix(const unsigned char* __restrict__ bases, const uint64_t* __restrict__ offs,
                                                  const uint64_t* __restrict__ starts, uint64_t n_reads, unsigned long long* codes, unsigned long long* bad,
                                                  const unsigned char* __restrict__ readflag, unsigned long long* max_len, uint64_t total,
                                                  unsigned long long* error_flags) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && offs[n_reads] - offs[0] != total) atomicOr(error_flags, 32ULL);     // fgpu_reads.total_bases was not what the offsets say
    {   // longest read of the batch: one look at the running maximum per BLOCK, an atomic only when the block raises it (loads of
        // one address from every wave queue up in its L2 channel just like same-address atomics do)
        __shared__ unsigned long long wave_max[4];
        unsigned long long l = i < n_reads ? (unsigned long long)(offs[i + 1] - offs[i]) : 0;
        for (int o = 32; o > 0; o >>= 1) { unsigned long long t = __shfl_down(l, o, 64); l = t > l ? t : l; }
        if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = l;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long m = wave_max[0];
            for (int w = 1; w < 4; w++) m = wave_max[w] > m ? wave_max[w] : m;
            if (m > *(volatile unsigned long long*)max_len) atomicMax(max_len, m);
        }
    }
    if (i >= n_reads || !readflag[i]) return;
    const uint64_t off0 = offs[0];
    const uint64_t L = offs[i + 1] - offs[i];
    const uint64_t S = (offs[i] - off0) + i;
    // the buffers were sized from the caller's total_bases: a read that ends beyond it is left alone -- the mismatch is the error reported above
    // (flag 32), never a write past the code and mask arrays
    if (S + L > total + n_reads) return;
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

// fgpu_create touches one kernel of every translation unit from a helper thread: the runtime loads a unit's code object at the first use of
// one of its kernels (20-25 ms for the large units), which otherwise lands on the first batch of each pass
void fgpu_touch_pack() {
    hipFuncAttributes attr;
    (void)hipFuncGetAttributes(&attr, (const void*)k_pack_fix);
}

// end of an API call that packed a host batch: everything that reads its staging set has been queued on the main stream
int fgpu_host_batch_done(fgpu_ctx* ctx, const fgpu_reads* reads) {
    if (reads->on_device || ctx->host_set < 0) return FGPU_OK;
    if (!ctx->ev_stage_free[ctx->host_set]) FGPU_HIP(hipEventCreateWithFlags(&ctx->ev_stage_free[ctx->host_set], hipEventDisableTiming));
    FGPU_HIP(hipEventRecord(ctx->ev_stage_free[ctx->host_set], ctx->stream));
    ctx->host_stage_used[ctx->host_set] = true;
    ctx->host_set = -1;
    return FGPU_OK;
}

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
        if (reads->offsets == ctx->split_offsets && n == ctx->split_n) {   // cut by fgpu_text_split just now: the total came with it
            total = ctx->split_total;
            // ... and so did the size of its chunk: the buffers of this batch that have to grow are made for the largest chunk announced
            ctx->ensure_scale = ctx->text_reserve > ctx->text_last_bytes && ctx->text_last_bytes
                                    ? std::min(8.0, (double)ctx->text_reserve / (double)ctx->text_last_bytes) : 1.0;
        } else if (reads->total_bases) {   // the caller's word for it; k_pack_fix compares it with the offsets (error flag 32)
            ctx->ensure_scale = 1.0;
            total = reads->total_bases;
        } else {
            uint64_t ends[2];
            FGPU_HIP(hipMemcpyAsync(&ends[0], reads->offsets, 8, hipMemcpyDeviceToHost, ctx->stream));
            FGPU_HIP(hipMemcpyAsync(&ends[1], reads->offsets + n, 8, hipMemcpyDeviceToHost, ctx->stream));
            FGPU_HIP(fgpu_sync_stream(ctx, ctx->stream));
            total = ends[1] - ends[0];
        }
        d_bases = (const unsigned char*)reads->bases;
        d_offs = reads->offsets;
    } else {
        ctx->ensure_scale = 1.0;
        total = reads->offsets[n] - reads->offsets[0];
        // Host buffers go through one of two staging sets on the text stream: the copy of this batch runs beside the kernels of the batch
        // before it, and the host waits for the copy only (so the caller may reuse its buffers) -- on the main stream the device idled
        // during every copy and the host waited for every batch.  A set is written again only after the call that used it has run
        // (ev_stage_free, recorded by fgpu_host_batch_done at the end of that call).
        int rc = fgpu_text_streams(ctx);
        if (rc) return rc;
        if (ctx->host_set >= 0) {   // the call that packed the previous host batch failed half-way: nothing is known about its set
            FGPU_HIP(fgpu_sync_stream(ctx, ctx->stream));
            ctx->host_stage_used[0] = ctx->host_stage_used[1] = false;
            ctx->host_set = -1;
        }
        const int b = (int)(ctx->host_turn++ & 1);
        ctx->host_set = b;
        DevBuf& sb = ctx->host_stage[2 * b];
        DevBuf& so = ctx->host_stage[2 * b + 1];
        if ((rc = fgpu_ensure(ctx, &sb, total + 16))) return rc;
        if ((rc = fgpu_ensure(ctx, &so, (n + 1) * 8))) return rc;
        if (ctx->host_stage_used[b]) FGPU_HIP(hipStreamWaitEvent(ctx->tstream, ctx->ev_stage_free[b], 0));
        // bases are copied from offsets[0] on, so the device copy is addressed with the same offsets
        // shifted by offsets[0]: keep the original offsets and bias the base pointer instead.
        FGPU_HIP(hipMemcpyAsync(sb.p, reads->bases + reads->offsets[0], total, hipMemcpyHostToDevice, ctx->tstream));
        FGPU_HIP(hipMemcpyAsync(so.p, reads->offsets, (n + 1) * 8, hipMemcpyHostToDevice, ctx->tstream));
        FGPU_HIP(hipEventRecord(ctx->ev_text_done, ctx->tstream));
        FGPU_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_text_done, 0));
        FGPU_HIP(fgpu_sync_stream(ctx, ctx->tstream));
        d_bases = (const unsigned char*)sb.p - reads->offsets[0];
        d_offs = (const uint64_t*)so.p;
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
        if ((rc = fgpu_ensure_b(ctx, &bb.codes, 64))) return rc;
        if ((rc = fgpu_ensure_b(ctx, &bb.bad, 64))) return rc;
        FGPU_HIP(hipMemsetAsync(bb.bad.p, 0xFF, 64, ctx->stream));
        FGPU_HIP(hipMemsetAsync(bb.codes.p, 0, 64, ctx->stream));
        return FGPU_OK;
    }
    if ((rc = fgpu_ensure_b(ctx, &bb.codes, (2 * (bb.n_words + FGPU_PADW)) * 8))) return rc;
    if ((rc = fgpu_ensure_b(ctx, &bb.bad, (bb.n_words + FGPU_PADW) * 8))) return rc;
    if ((rc = fgpu_ensure_b(ctx, &bb.readflag, n + 16))) return rc;
    FGPU_HIP(hipMemsetAsync(bb.readflag.p, 0, n, ctx->stream));
    // four trips of 64 x 32 positions per wave: the bisection that opens a wave's run is paid once per 8192 positions
    FGPU_LAUNCH("pack", k_pack, fgpu_grid((bb.n_words + FGPU_PADW) / 2 + 1, 256), 256, d_bases, d_offs, d_starts, n, T, bb.n_words, (uint64_t*)bb.codes.p,
                (uint32_t*)bb.bad.p, (unsigned char*)bb.readflag.p);
    FGPU_LAUNCH("pack_fix", k_pack_fix, fgpu_blocks(n, 256), 256, d_bases, d_offs, d_starts, n, (unsigned long long*)bb.codes.p,
                (unsigned long long*)bb.bad.p, (const unsigned char*)bb.readflag.p, &ctx->counters->max_read_len, total, &ctx->counters->error_flags);
    return FGPU_OK;
}
