// text.hip — input side: FASTA / FASTQ record splitting on the device.
//
// Replaces the reading loops of the reference (utils/Bloom.cpp:280-282,340; src/ReadScanner.cpp:306-308,349):
//     while (getline(header)) { getline(sequence); ...; if (fastq) { getline; getline; } }
// A record is P = 2 (FASTA) or 4 (FASTQ) lines; its read is line 1 of the record, whatever the lines contain (the
// reference never looks at '>' / '@' / '+').  On a chunk of raw file text:
//   k_text_newlines   one lane per byte, ballot -> newline bit plane + per-word counts
//   rocPRIM scan      rank of every newline
//   k_text_records    every newline of rank m: m % P == 0 ends a header (the read starts behind it), m % P == 1 ends a read,
//                     m % P == P-1 ends a record
//   rocPRIM scan      lengths -> offsets
// The batch handed out points INTO the text (fgpu_reads.starts): no compaction copy; pack.hip reads the bases from there.
// Streaming (1 byte read per byte of text), negligible next to the passes; it exists so that a host does not spend
// seconds in getline per pass while the device needs a tenth of that for the pass itself.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "fgpu_ctx.h"

namespace {

// 0x80 in every byte of x that equals '\n' (exact: no borrow between bytes), gathered into 8 bits, lowest byte first
__device__ __forceinline__ uint32_t nl_mask8(uint64_t x) {
    const uint64_t t = x ^ 0x0A0A0A0A0A0A0A0AULL;                                                     // zero bytes <=> newlines
    const uint64_t y = ~(((t & 0x7F7F7F7F7F7F7F7FULL) + 0x7F7F7F7F7F7F7F7FULL) | t | 0x7F7F7F7F7F7F7F7FULL);   // 0x80 where the byte is zero
    return (uint32_t)(((y >> 7) * 0x0102040810204080ULL) >> 56);
}

// One lane per 64 bytes of text = one word of the newline plane: four 16-byte loads, the byte compares as SWAR arithmetic.  (Round 6: one lane
// per BYTE and a ballot -- the kernel of rounds 1-5 -- ran at 41 GB/s beside the pass's kernels, 0.38 s of config 5's 1.4 s of passes through the
// command line: profiles/r06_cli_large.txt.)  `aligned`: text is 16-byte aligned and readable up to n_words * 64 (the library's own copy of a
// host chunk); else -- a caller's device buffer -- bytes are read one by one and never past n.
__global__ void __launch_bounds__(256) k_text_newlines(const unsigned char* __restrict__ text, uint64_t n, uint64_t n_words,
                                                       uint64_t* __restrict__ nl, uint32_t* __restrict__ count, int aligned) {
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t m = 0;
        if (aligned) {
            const uint4* p = (const uint4*)(text + w * 64);
            const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
            m = (uint64_t)nl_mask8((uint64_t)a.x | ((uint64_t)a.y << 32)) | ((uint64_t)nl_mask8((uint64_t)a.z | ((uint64_t)a.w << 32)) << 8) |
                ((uint64_t)nl_mask8((uint64_t)b.x | ((uint64_t)b.y << 32)) << 16) | ((uint64_t)nl_mask8((uint64_t)b.z | ((uint64_t)b.w << 32)) << 24) |
                ((uint64_t)nl_mask8((uint64_t)c.x | ((uint64_t)c.y << 32)) << 32) | ((uint64_t)nl_mask8((uint64_t)c.z | ((uint64_t)c.w << 32)) << 40) |
                ((uint64_t)nl_mask8((uint64_t)d.x | ((uint64_t)d.y << 32)) << 48) | ((uint64_t)nl_mask8((uint64_t)d.z | ((uint64_t)d.w << 32)) << 56);
            if ((w + 1) * 64 > n) m &= (1ULL << (n - w * 64)) - 1;            // (the bytes behind the text are whatever the buffer held)
        } else {
            const uint64_t lo = w * 64, hi = lo + 64 < n ? lo + 64 : n;
            for (uint64_t q = lo; q < hi; q++) m |= (uint64_t)(text[q] == '\n') << (q - lo);
        }
        nl[w] = m;
        count[w] = (uint32_t)__popcll(m);
    }
}

// defaults for records whose header / read line is not terminated inside the text (only possible in the final chunk):
// an absent read is empty, an unterminated one runs to the end of the text
__global__ void __launch_bounds__(256) k_text_defaults(uint64_t* __restrict__ starts, uint64_t* __restrict__ ends, uint64_t n_rec, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rec; i += (uint64_t)gridDim.x * blockDim.x) {
        starts[i] = n;
        ends[i] = n;
    }
}

__global__ void __launch_bounds__(256) k_text_records(const uint64_t* __restrict__ nl, const uint32_t* __restrict__ rank, uint64_t n_words,
                                                      uint32_t P, uint64_t n_rec, uint64_t* __restrict__ starts, uint64_t* __restrict__ ends,
                                                      uint64_t* __restrict__ rec_end) {
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t bits = nl[w];
        uint64_t m = rank[w];
        while (bits) {
            const uint64_t pos = w * 64 + __builtin_ctzll(bits);
            bits &= bits - 1;
            const uint64_t r = m / P, c = m % P;
            if (r < n_rec) {
                if (c == 0) starts[r] = pos + 1;
                else if (c == 1) ends[r] = pos;
                if (c == P - 1) rec_end[r] = pos + 1;
            }
            m++;
        }
    }
}

__global__ void __launch_bounds__(256) k_text_lengths(const uint64_t* __restrict__ starts, const uint64_t* __restrict__ ends, uint64_t n_rec,
                                                      uint64_t* __restrict__ len) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n_rec; i += (uint64_t)gridDim.x * blockDim.x)
        len[i] = i < n_rec ? ends[i] - starts[i] : 0;   // one extra 0 so that the exclusive scan also yields the total
}

}  // namespace

// fgpu_create touches one kernel of every translation unit from a helper thread: the runtime loads a unit's code object at the first use of
// one of its kernels (20-25 ms for the large units), which otherwise lands on the first batch of each pass
void fgpu_touch_text() {
    hipFuncAttributes attr;
    (void)hipFuncGetAttributes(&attr, (const void*)k_text_lengths);
}

// The split runs on its own stream with three sets of buffers used in turn (two until round 6): while the main stream works on the batch cut out of the previous
// chunk, this chunk is copied to the device and cut, and the host waits for the text stream only.  (On the main stream -- round 1 -- every
// chunk waited for the batch before it, the device idled during the copy, and a 1.1 GB file took 2.4 times the kernels' time per pass.)
int fgpu_text_streams(fgpu_ctx* ctx) {
    if (ctx->tstream) return FGPU_OK;
    for (int i = 0; i < FGPU_TEXT_SETS; i++)
        if (!ctx->ev_text_mark[i]) FGPU_HIP(hipEventCreateWithFlags(&ctx->ev_text_mark[i], hipEventDisableTiming));
    if (!ctx->ev_text_done) FGPU_HIP(hipEventCreateWithFlags(&ctx->ev_text_done, hipEventDisableTiming));
    // High priority (round 6): the host WAITS for this stream before it can queue the batch, and the pass's kernels on the main stream are fixed
    // grids of long-lived blocks -- at equal priority the newline kernel's blocks are dispatched as those retire, 0.9 ms per 64 MB chunk where
    // the kernel alone needs 0.03 (rocprofv3 of `faucet` on config 4's 22 GB: 598 ms in 666 launches, beside the pass, not in front of it)
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);   // hi = numerically lowest = highest priority
    FGPU_HIP(hipStreamCreateWithPriority(&ctx->tstream, hipStreamNonBlocking, hi));   // last: its presence says the events are there
    return FGPU_OK;
}

int fgpu_text_reserve(fgpu_ctx* ctx, uint64_t max_chunk_bytes) {
    if (!ctx) return FGPU_ERR_ARG;
    ctx->text_reserve = max_chunk_bytes;          // a hint: the next fgpu_text_split sizes its buffers for it (larger chunks still work)
    return FGPU_OK;
}

extern "C" int fgpu_text_split(fgpu_ctx* ctx, const char* text, uint64_t nbytes, int text_on_device, int fastq, int final_chunk,
                               fgpu_reads* out, uint64_t* consumed) {
    if (!ctx || !out || !consumed || (nbytes && !text)) return FGPU_ERR_ARG;
    // (prepared scan batches -- fgpu_scan_prepare, read shards -- do not point into the text any more: a batch is packed into its own code and
    // plane buffers inside the call that takes it, and that call ends with a synchronisation of the main stream, the piece count)
    FGPU_HIP(hipSetDevice(ctx->prm.device));
    memset(out, 0, sizeof(*out));
    out->on_device = 1;
    *consumed = 0;
    ctx->split_offsets = nullptr;
    ctx->text_last_bytes = nbytes;
    if (nbytes == 0) return FGPU_OK;
    int rc;
    if ((rc = fgpu_text_streams(ctx))) return rc;
    // Set c % 3 was last read by the batch of call c - 3, which was queued before call c - 2 began: the mark call c - 2 left on the main
    // stream -- so this chunk is cut while the batches of calls c - 2 and c - 1 run, and the main stream has a whole batch queued behind the one
    // it works on.  Text that is on the device already was written in main-stream order by the caller: this call's own mark.
    const uint64_t c = ctx->text_calls++;
    TextSet& ts = ctx->text[c % FGPU_TEXT_SETS];
    FGPU_HIP(hipEventRecord(ctx->ev_text_mark[c % FGPU_TEXT_SETS], ctx->stream));
    if (text_on_device) FGPU_HIP(hipStreamWaitEvent(ctx->tstream, ctx->ev_text_mark[c % FGPU_TEXT_SETS], 0));
    else if (c >= FGPU_TEXT_SETS - 1) FGPU_HIP(hipStreamWaitEvent(ctx->tstream, ctx->ev_text_mark[(c - (FGPU_TEXT_SETS - 1)) % FGPU_TEXT_SETS], 0));
    struct OnStream {   // kernels of this call go to the text stream
        fgpu_ctx* c;
        hipStream_t saved;
        OnStream(fgpu_ctx* ctx) : c(ctx), saved(ctx->launch_stream) { ctx->launch_stream = ctx->tstream; }
        ~OnStream() { c->launch_stream = saved; }
    } on_stream(ctx);
    hipStream_t st = ctx->tstream;
    // The buffers of a set are sized for the largest chunk the caller has announced (fgpu_text_reserve), not for this one: a caller that
    // starts with small chunks and doubles them (the command line: 1/4, 1/4, 1/2, 1 of -chunk_mb) made every buffer of both sets be freed
    // and allocated again three times, each a synchronisation of the device -- 20 ms of a 77 ms pass over a 1 GB file.
    const uint64_t cap = std::max<uint64_t>(nbytes, ctx->text_reserve);
    const uint64_t cap_words = (cap + 63) / 64;
    const unsigned char* d_text;
    if (text_on_device) {
        d_text = (const unsigned char*)text;
    } else {
        if ((rc = fgpu_ensure(ctx, &ts.buf, cap + 64))) return rc;
        FGPU_HIP(hipMemcpyAsync(ts.buf.p, text, nbytes, hipMemcpyHostToDevice, st));
        d_text = (const unsigned char*)ts.buf.p;
    }
    const uint64_t n_words = (nbytes + 63) / 64;
    const uint32_t P = fastq ? 4u : 2u;
    if ((rc = fgpu_ensure(ctx, &ts.nl, cap_words * 8))) return rc;
    if ((rc = fgpu_ensure(ctx, &ts.rank, (2 * cap_words + 2) * 4))) return rc;
    uint64_t* nl = (uint64_t*)ts.nl.p;
    uint32_t* count = (uint32_t*)ts.rank.p;
    uint32_t* rank = count + n_words + 1;
    // (the library's own copy of a host chunk is 256-byte aligned and cap + 64 bytes long; a caller's device text is taken byte by byte unless aligned AND not the tail)
    const int aligned = (!text_on_device && ((uintptr_t)d_text & 15) == 0) ? 1 : 0;
    FGPU_LAUNCH("text_newlines", k_text_newlines, fgpu_grid(n_words, 256), 256, d_text, nbytes, n_words, nl, count, aligned);
    size_t tmp_bytes = 0, tmp2 = 0;
    FGPU_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, count, rank, 0u, n_words, rocprim::plus<uint32_t>(), st));
    FGPU_HIP(rocprim::exclusive_scan(nullptr, tmp2, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint64_t)0, n_words * 64 / 2 + 2, rocprim::plus<uint64_t>(), st));
    if (tmp2 > tmp_bytes) tmp_bytes = tmp2;   // the second scan (at most one record per 2 bytes) reuses the scratch
    size_t tmp_cap = tmp_bytes;
    if (cap_words > n_words) {                // (the scratch the largest announced chunk will ask for)
        size_t a = 0, b = 0;
        FGPU_HIP(rocprim::exclusive_scan(nullptr, a, count, rank, 0u, cap_words, rocprim::plus<uint32_t>(), st));
        FGPU_HIP(rocprim::exclusive_scan(nullptr, b, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint64_t)0, cap_words * 64 / 2 + 2, rocprim::plus<uint64_t>(), st));
        tmp_cap = std::max(tmp_cap, std::max(a, b));
    }
    if ((rc = fgpu_ensure(ctx, &ts.tmp, tmp_cap + 16))) return rc;
    FGPU_HIP(rocprim::exclusive_scan(ts.tmp.p, tmp_bytes, count, rank, 0u, n_words, rocprim::plus<uint32_t>(), st));
    uint32_t last[2];
    unsigned char last_byte = 0;
    FGPU_HIP(hipMemcpyAsync(&last[0], count + n_words - 1, 4, hipMemcpyDeviceToHost, st));
    FGPU_HIP(hipMemcpyAsync(&last[1], rank + n_words - 1, 4, hipMemcpyDeviceToHost, st));
    FGPU_HIP(hipMemcpyAsync(&last_byte, d_text + nbytes - 1, 1, hipMemcpyDeviceToHost, st));
    FGPU_HIP(fgpu_sync_stream(ctx, st));
    const uint64_t n_newlines = (uint64_t)last[0] + last[1];
    // lines as getline counts them: every '\n' ends one; at the end of the file a non-empty unterminated tail is one more
    const uint64_t n_lines = n_newlines + ((final_chunk && last_byte != '\n') ? 1 : 0);
    const uint64_t n_rec = final_chunk ? (n_lines + P - 1) / P : n_newlines / P;
    if (n_rec == 0) return FGPU_OK;   // not even one complete record: the caller reads more
    if (n_rec >= 0xFFFFFFFFULL) { ctx->err = "more than 2^32 records in one chunk of text"; return FGPU_ERR_CAPACITY; }
    if ((rc = fgpu_ensure(ctx, &ts.rec, (4 * std::max<uint64_t>(n_rec, n_rec * cap / nbytes + 16) + 4) * 8))) return rc;   // (records of a chunk of the reserved size, at this chunk's density)
    uint64_t* starts = (uint64_t*)ts.rec.p;
    uint64_t* ends = starts + n_rec;         // becomes the lengths
    uint64_t* offsets = ends + n_rec + 1;    // n_rec + 1 entries
    uint64_t* rec_end = offsets + n_rec + 1;
    FGPU_LAUNCH("text_defaults", k_text_defaults, fgpu_grid(n_rec, 256), 256, starts, ends, n_rec, nbytes);
    FGPU_HIP(hipMemsetAsync(rec_end, 0, n_rec * 8, st));
    FGPU_LAUNCH("text_records", k_text_records, fgpu_grid(n_words, 256), 256, (const uint64_t*)nl, (const uint32_t*)rank, n_words, P, n_rec,
                starts, ends, rec_end);
    FGPU_LAUNCH("text_lengths", k_text_lengths, fgpu_grid(n_rec + 1, 256), 256, (const uint64_t*)starts, (const uint64_t*)ends, n_rec, ends);
    size_t need = tmp_bytes;
    FGPU_HIP(rocprim::exclusive_scan(ts.tmp.p, need, ends, offsets, (uint64_t)0, n_rec + 1, rocprim::plus<uint64_t>(), st));
    uint64_t used = nbytes, total = 0;
    if (!final_chunk) FGPU_HIP(hipMemcpyAsync(&used, rec_end + n_rec - 1, 8, hipMemcpyDeviceToHost, st));
    FGPU_HIP(hipMemcpyAsync(&total, offsets + n_rec, 8, hipMemcpyDeviceToHost, st));   // saves the batch call its own look at the offsets
    FGPU_HIP(hipEventRecord(ctx->ev_text_done, st));
    FGPU_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_text_done, 0));
    FGPU_HIP(fgpu_sync_stream(ctx, st));
    *consumed = used;
    out->bases = (const char*)d_text;
    out->offsets = offsets;
    out->starts = starts;
    out->n_reads = n_rec;
    ctx->split_offsets = offsets;
    ctx->split_n = n_rec;
    ctx->split_total = total;
    return FGPU_OK;
}

extern "C" void* fgpu_host_alloc(uint64_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}

extern "C" void fgpu_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}
