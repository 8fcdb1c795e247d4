// stage3.hip — JunctionMap::findNeighbor (utils/JunctionMap.cpp:231-412) for many (junction, extension) pairs at once, whole walks on the
// device.
//
// The reference's contig-graph stage (not part of this build) starts a walk at every junction along every extension it builds a contig on:
// getValidJExtension (:474-490; 4 x (Bloom::oldContains + JChecker::jcheck)) names the one next base, the DoubleKmer advances, the junction
// map is asked twice per step (backward- and forward-facing half step), until a junction or a sink is reached.  One call is a chain of
// dependent filter probes; different calls are independent -- one lane per walk.  The junction map the walks look into is handed over by the
// caller (fgpu_stage3_set_junctions: the scan's own result, or a parsed .junctions file) and kept as an open-addressing table in HBM: k-mer
// -> the five distances, which is all findNeighbor reads of a Junction.
//
// The control flow is the one of faucet_amd/stage3.py (findNeighbor in lock-step over fgpu_probe_valid_extension, one device call per step,
// bookkeeping in numpy), which is pinned on the reference's own findNeighbor (tests/golden/stage3_neighbors_*.jsonl.gz); this kernel is
// checked against both.  No filter bit and no map entry is looked at on the host.
#include <hip/hip_runtime.h>

#include "fgpu_ctx.h"
#include "fgpu_flags.h"

namespace {

constexpr uint64_t S3_EMPTY = ~0ULL;

__device__ __forceinline__ uint64_t s3_slot(uint64_t key, uint64_t mask) { return (key * 0x9E3779B97F4A7C15ULL >> 20) & mask; }

// one junction per thread: key -> packed distances (byte i = dist[i])
__global__ void __launch_bounds__(256) k_s3_build(const uint64_t* __restrict__ keys, const fgpu_junction* __restrict__ recs, uint64_t n,
                                                  unsigned long long* __restrict__ tkeys, uint64_t* __restrict__ tdist, uint64_t mask,
                                                  unsigned int* __restrict__ repeated) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t key = keys[i];
    uint64_t d = 0;
    for (int b = 0; b < 5; b++) d |= (uint64_t)recs[i].dist[b] << (8 * b);
    uint64_t s = s3_slot(key, mask);
    for (uint64_t probe = 0; probe <= mask; probe++) {
        const unsigned long long old = atomicCAS(&tkeys[s], (unsigned long long)S3_EMPTY, (unsigned long long)key);
        if (old == S3_EMPTY) { tdist[s] = d; return; }
        if (old == key) { atomicAdd(repeated, 1u); return; }
        s = (s + 1) & mask;
    }
}

// junctionMap.find(kmer): the packed distances, or false
__device__ __forceinline__ bool s3_find(const uint64_t* __restrict__ tkeys, const uint64_t* __restrict__ tdist, uint64_t mask, uint64_t key, uint64_t& dist) {
    uint64_t s = s3_slot(key, mask);
    for (uint64_t probe = 0; probe <= mask; probe++) {
        const uint64_t k = tkeys[s];
        if (k == key) { dist = tdist[s]; return true; }
        if (k == S3_EMPTY) return false;
        s = (s + 1) & mask;
    }
    return false;
}
__device__ __forceinline__ int s3_dist(uint64_t packed, int i) { return (int)((packed >> (8 * i)) & 0xFF); }

// JunctionMap::getValidJExtension (utils/JunctionMap.cpp:474-490): -1 none, -2 several, else the nucleotide
__device__ int s3_valid_extension(uint64_t km, const FdParams& fp, const uint32_t* __restrict__ bloom) {
    int r = -1;
    for (int nt = 0; nt < 4; nt++) {
        const uint64_t e = ((km << 2) | (uint64_t)nt) & fp.kmask;
        if (fd_bloom_contains_canon(bloom, fd_canon(e, fp.k), fp.tai_mask, fp.n_hash) && jcheck_dfs(e, fp, bloom)) {
            if (r != -1) return -2;
            r = nt;
        }
    }
    return r;
}

__global__ void __launch_bounds__(256) k_s3_find_neighbors(const uint64_t* __restrict__ starts, const signed char* __restrict__ indices, uint64_t n,
                                                           FdParams fp, const uint32_t* __restrict__ bloom, const uint64_t* __restrict__ tkeys,
                                                           const uint64_t* __restrict__ tdist, uint64_t mask, int max_read_length,
                                                           fgpu_neighbor* __restrict__ out, unsigned long long* __restrict__ n_probes,
                                                           uint64_t* __restrict__ contigs, uint64_t stride) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long probes = 0;
    if (w < n) {
        const int k = fp.k, idx = indices[w];
        uint64_t km = starts[w] & fp.kmask, rc = fd_revcomp(km, k);
        fgpu_neighbor res;
        res.kmer = 0; res.dist = 0; res.len = 0; res.node = 0; res.rindex = 0; res.abort = 0; res.reserved = 0; res.reserved2 = 0;
        fgpu_neighbor sink = res;
        uint64_t packed = 0;
        bool done = false;
        if (idx < 0 || idx > 4 || !s3_find(tkeys, tdist, mask, km, packed)) {
            res.abort = 2;                                  // not a junction of the map, or no such extension: the caller's mistake, not the reference's
            done = true;
        }
        const int maxd = done ? 0 : s3_dist(packed, idx);
        int dist = 1, ln = 0, lastnuc = 0, retidx = 4, phase = 0;
        auto forward = [&](int nuc) {                        // DoubleKmer::forward (utils/DoubleKmer.cpp:17-23)
            km = ((km << 2) | (uint64_t)nuc) & fp.kmask;
            rc = (rc >> 2) | ((uint64_t)(nuc ^ 2) << (2 * k - 2));
        };
        auto swap = [&]() { const uint64_t t = km; km = rc; rc = t; };
        // the contig string of the reference's result (:241,254-256,272-276,343), 2 bits per base, 32 bases per word, first base lowest
        uint64_t* const text = contigs ? contigs + w * stride : nullptr;
        uint64_t acc = 0;
        uint32_t pos = 0;
        auto emit = [&](int code) {
            if (!text) return;
            acc |= (uint64_t)code << (2 * (pos & 31));
            pos++;
            if ((pos & 31) == 0) {
                if ((pos >> 5) <= stride) text[(pos >> 5) - 1] = acc;
                acc = 0;
            }
        };
        auto emit_kmer = [&](uint64_t x) { for (int i = k - 1; i >= 0; i--) emit((int)((x >> (2 * i)) & 3)); };
        auto finish = [&](uint64_t kmer, int node, int rindex, int d, int len) {
            res.kmer = kmer; res.node = (int8_t)node; res.rindex = (int8_t)rindex; res.dist = d; res.len = len;
            done = true;
        };
        uint64_t other = 0;
        if (!done) {
            // ---- the first one or two k-mers (:251-302)
            if (idx == 4) {
                swap();                                     // doubleKmer.reverse()
                ln = k;
                emit_kmer(km);
                if (s3_find(tkeys, tdist, mask, km, other)) finish(km, 1, 4, 1, ln);
            } else {
                lastnuc = (int)(rc & 3);
                emit((int)((km >> (2 * (k - 1))) & 3));
                forward(idx);
                ln = 1 + k;
                emit_kmer(km);
                if (s3_find(tkeys, tdist, mask, rc, other)) finish(rc, 1, lastnuc, 1, ln);
                else if (maxd == 1) finish(rc, 0, lastnuc, 1, ln);
                else {
                    dist = 2;
                    if (s3_find(tkeys, tdist, mask, km, other)) finish(km, 1, 4, 2, ln);
                }
            }
            if (!done && dist > maxd) { res.abort = 1; done = true; }   // the reference's assert(dist <= maxDist)
        }
        while (!done) {
            // ---- leaving the first loop (:343-366): a junction exactly where expected, else this is (probably) a sink
            if (phase == 0 && dist >= maxd) {
                if (s3_find(tkeys, tdist, mask, km, other)) { finish(km, 1, retidx, dist, ln); break; }
                sink.kmer = km; sink.node = 0; sink.rindex = 4; sink.dist = dist; sink.len = ln;
                phase = 1;
            }
            if (phase == 1 && dist >= maxd + 2 * max_read_length) { res = sink; break; }   // :376, ran past every possible overlap
            const int ext = s3_valid_extension(km, fp, bloom);
            probes++;
            if (ext < 0) {
                if (phase == 0) res.abort = 1;              // assert(validExtension != -1 / -2)
                else res = sink;                            // off the real sequence: the sink stands
                break;
            }
            lastnuc = (int)(rc & 3);
            forward(ext);
            ln++;
            emit(ext);
            // backward-facing half step
            dist++;
            swap();
            retidx = lastnuc;
            bool isj = s3_find(tkeys, tdist, mask, km, other);
            if (phase == 0) {
                if (dist == maxd) continue;                 // break of the first loop: handled at the top of the next round
                if (isj) { finish(km, 1, retidx, dist, ln); break; }
            } else if (isj) {                               // past maxDist: overlap test (:391-404)
                if (s3_dist(other, lastnuc) + maxd - dist >= 0) finish(km, 1, lastnuc, dist, ln);
                else res = sink;
                break;
            }
            // forward-facing half step
            dist++;
            swap();
            retidx = 4;
            isj = s3_find(tkeys, tdist, mask, km, other);
            if (phase == 0) {
                if (dist == maxd) continue;
                if (isj) { finish(km, 1, 4, dist, ln); break; }
            } else if (isj) {
                if (s3_dist(other, 4) + maxd - dist >= 0) finish(km, 1, 4, dist, ln);
                else res = sink;
                break;
            }
        }
        out[w] = res;
        if (text && (pos & 31) && (pos >> 5) < stride) text[pos >> 5] = acc;
    }
    // one atomic per wave
    for (int off = 32; off; off >>= 1) probes += __shfl_down(probes, off);
    if ((threadIdx.x & 63) == 0 && probes) atomicAdd(n_probes, probes);
}

}  // namespace

extern "C" int fgpu_stage3_set_junctions(fgpu_ctx* ctx, const uint64_t* keys_host, const fgpu_junction* recs_host, uint64_t n) {
    if (!ctx || (n && (!keys_host || !recs_host))) return FGPU_ERR_ARG;
    if (ctx->phase != 0) { ctx->err = "fgpu_stage3_set_junctions while a pass is open"; return FGPU_ERR_STATE; }
    FGPU_HIP(hipSetDevice(ctx->prm.device));
    ctx->s3_ready = false;
    uint64_t cap = 1024;
    while (cap < 2 * n + 2) cap <<= 1;
    int rc;
    if ((rc = fgpu_ensure(ctx, &ctx->s3_keys, cap * 8))) return rc;
    if ((rc = fgpu_ensure(ctx, &ctx->s3_dist, cap * 8))) return rc;
    if ((rc = fgpu_ensure(ctx, &ctx->s3_in, n * (8 + sizeof(fgpu_junction)) + 64))) return rc;
    ctx->s3_mask = cap - 1;
    ctx->s3_count = 0;
    FGPU_HIP(hipMemsetAsync(ctx->s3_keys.p, 0xFF, cap * 8, ctx->stream));
    FGPU_HIP(hipMemsetAsync(ctx->s3_in.p, 0, 64, ctx->stream));
    if (n) {
        uint64_t* d_keys = (uint64_t*)((char*)ctx->s3_in.p + 64);
        fgpu_junction* d_recs = (fgpu_junction*)(d_keys + n);
        FGPU_HIP(hipMemcpyAsync(d_keys, keys_host, n * 8, hipMemcpyHostToDevice, ctx->stream));
        FGPU_HIP(hipMemcpyAsync(d_recs, recs_host, n * sizeof(fgpu_junction), hipMemcpyHostToDevice, ctx->stream));
        FGPU_LAUNCH("s3_build", k_s3_build, fgpu_blocks(n, 256), 256, (const uint64_t*)d_keys, (const fgpu_junction*)d_recs, n,
                    (unsigned long long*)ctx->s3_keys.p, (uint64_t*)ctx->s3_dist.p, ctx->s3_mask, (unsigned int*)ctx->s3_in.p);
    }
    unsigned int repeated = 0;
    FGPU_HIP(hipMemcpyAsync(&repeated, ctx->s3_in.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    FGPU_HIP(fgpu_sync_stream(ctx, ctx->stream));
    if (repeated) { ctx->err = "fgpu_stage3_set_junctions: a k-mer occurs more than once"; return FGPU_ERR_ARG; }
    ctx->s3_count = n;
    ctx->s3_ready = true;
    return FGPU_OK;
}

// longest contig string a walk can return: the start (k + 1 bases) and one base per step, two half steps of distance each, up to
// maxDist (a byte) + 2 * max_read_length (:376)
extern "C" uint64_t fgpu_stage3_contig_words(int32_t k, int32_t max_read_length) {
    const uint64_t bases = (uint64_t)k + 1 + (255 + 2 * (uint64_t)max_read_length) / 2 + 2;
    return (bases + 31) / 32;
}

extern "C" int fgpu_stage3_find_neighbors(fgpu_ctx* ctx, const uint64_t* start_kmers_host, const int8_t* indices_host, uint64_t n,
                                          int32_t max_read_length, fgpu_neighbor* out, uint64_t* n_probes, uint64_t* contigs_out,
                                          uint64_t contig_stride_words) {
    if (!ctx || (n && (!start_kmers_host || !indices_host || !out)) || max_read_length < 1) return FGPU_ERR_ARG;
    if (contigs_out && contig_stride_words < fgpu_stage3_contig_words(ctx->prm.k, max_read_length)) {
        ctx->err = "fgpu_stage3_find_neighbors: contig_stride_words is less than fgpu_stage3_contig_words(k, max_read_length)";
        return FGPU_ERR_ARG;
    }
    if (ctx->phase != 0) { ctx->err = "fgpu_stage3_find_neighbors while a pass is open"; return FGPU_ERR_STATE; }
    if (!ctx->s3_ready) { ctx->err = "fgpu_stage3_find_neighbors before fgpu_stage3_set_junctions"; return FGPU_ERR_STATE; }
    if (n_probes) *n_probes = 0;
    if (!n) return FGPU_OK;
    FGPU_HIP(hipSetDevice(ctx->prm.device));
    int rc;
    const uint64_t in_bytes = (n * 9 + 15) & ~15ULL;
    const uint64_t stride = contigs_out ? contig_stride_words : 0;
    if ((rc = fgpu_ensure(ctx, &ctx->probe_buf, 64 + in_bytes + n * sizeof(fgpu_neighbor) + n * stride * 8))) return rc;
    char* base = (char*)ctx->probe_buf.p;
    unsigned long long* d_probes = (unsigned long long*)base;
    uint64_t* d_starts = (uint64_t*)(base + 64);
    signed char* d_idx = (signed char*)(d_starts + n);
    fgpu_neighbor* d_out = (fgpu_neighbor*)(base + 64 + in_bytes);
    uint64_t* d_contigs = stride ? (uint64_t*)(d_out + n) : nullptr;
    FGPU_HIP(hipMemsetAsync(d_probes, 0, 8, ctx->stream));
    FGPU_HIP(hipMemcpyAsync(d_starts, start_kmers_host, n * 8, hipMemcpyHostToDevice, ctx->stream));
    FGPU_HIP(hipMemcpyAsync(d_idx, indices_host, n, hipMemcpyHostToDevice, ctx->stream));
    FGPU_LAUNCH("s3_find_neighbors", k_s3_find_neighbors, fgpu_blocks(n, 256), 256, (const uint64_t*)d_starts, (const signed char*)d_idx, n, ctx->fd,
                (const uint32_t*)ctx->bloo2, (const uint64_t*)ctx->s3_keys.p, (const uint64_t*)ctx->s3_dist.p, ctx->s3_mask, (int)max_read_length, d_out,
                d_probes, d_contigs, stride);
    unsigned long long probes = 0;
    FGPU_HIP(hipMemcpyAsync(out, d_out, n * sizeof(fgpu_neighbor), hipMemcpyDeviceToHost, ctx->stream));
    FGPU_HIP(hipMemcpyAsync(&probes, d_probes, 8, hipMemcpyDeviceToHost, ctx->stream));
    if (stride) FGPU_HIP(hipMemcpyAsync(contigs_out, d_contigs, n * stride * 8, hipMemcpyDeviceToHost, ctx->stream));
    FGPU_HIP(fgpu_sync_stream(ctx, ctx->stream));
    if (n_probes) *n_probes = probes;
    return FGPU_OK;
}
