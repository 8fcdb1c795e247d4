// scan_pure.hip — pass 2, the part that is a pure function of (reads, bloo2).
//
// Replaces, for every position of the batch at once:
//   ReadScanner::getValidReads   (src/ReadScanner.cpp:233-257)  one oldContains(canon) per window; maximal runs of
//                                >= k present windows become "valid pieces" (the strings scan_forward walks)
//   scanInputRead's length gate  (src/ReadScanner.cpp:268)      segment length >= k + 2j + 1
//   ReadScanner::testForJunction (src/ReadScanner.cpp:36-56)    for nt != real extension: if oldContains(canon(ext))
//                                { NbJCheckKmer++; if jcheck(ext) return true; }
//   JChecker::jcheck(kmer_type)  (utils/JChecker.cpp:51-80)     is there a chain of j forward extensions in the filter
//
// Outputs, one bit per stream position (LSB-first words, written with one ballot per wave = coalesced 8-byte stores):
//   valid  window at p is in bloo2            pm   window p belongs to a valid piece      ps   p is the first window of a piece
//   ff/fb  testForJunction at (p, FORWARD) / (p, BACKWARD)
//   cf0,cf1 / cb0,cb1  two bit-planes of the NbJCheckKmer increment (0..3) at (p, FORWARD) / (p, BACKWARD)
// plus the piece list {start, windows} in stream (= processing) order.
//
// Roofline: random bit probes into bloo2 (Infinity-Cache / HBM); algorithmic bytes per k-mer:
//   L/(L-k+1) B of bases + 64 B per bit test the reference semantics perform (validity, alternate extensions, jcheck).
#include <algorithm>

#include "fgpu_ctx.h"
#include "fgpu_flags.h"

namespace {

__device__ __forceinline__ void wave_add(unsigned long long* dst, unsigned long long v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (fd_lane() == 0 && v) atomicAdd(dst, v);
}

// one atomic per BLOCK of 256 threads (see load.hip); every thread of the block must call
__device__ __forceinline__ void block_add(unsigned long long* dst, unsigned long long v) {
    __shared__ unsigned long long part[4];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (fd_lane() == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t = part[0] + part[1] + part[2] + part[3];
        if (t) atomicAdd(dst, t);
    }
    __syncthreads();
}

// Is this batch, word for word, the load batch kept under the same index?  *same starts non-zero; any differing word clears it.
__global__ void __launch_bounds__(256) k_scan_same(const uint64_t* __restrict__ codes, const uint64_t* __restrict__ kept_codes, uint64_t n_code_words,
                                                   const uint64_t* __restrict__ bad, const uint64_t* __restrict__ kept_bad, uint64_t n_words,
                                                   uint32_t* same) {
    bool differ = false;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_code_words; i += (uint64_t)gridDim.x * blockDim.x) {
        differ |= codes[i] != kept_codes[i];
        if (i < n_words) differ |= bad[i] != kept_bad[i];
    }
    if (__ballot(differ) && fd_lane() == 0) *same = 0;
}

// fixed grid striding over the stream, lanes = consecutive positions; counters live in registers until the wave retires.
// sure/same (optional): occurrences the load pass routed to bloo2 are in the filter by construction -- when the batch is the
// load batch (same != 0) their answer is "present" without a probe.
__global__ void __launch_bounds__(256) k_scan_valid(const uint64_t* __restrict__ codes, const uint64_t* __restrict__ bad,
                                                    uint64_t T, uint64_t n_words, FdParams fp, const uint32_t* __restrict__ bloom,
                                                    const uint64_t* __restrict__ sure, const uint32_t* __restrict__ same,
                                                    uint64_t* __restrict__ valid, DevCounters* cnt) {
    unsigned long long n_ok = 0, n_reused = 0;
    const uint64_t total = n_words * 64;
    const bool reuse = sure && *same != 0;
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (uint64_t)gridDim.x * blockDim.x) {
        bool ok = p < T && fd_window_ok(bad, p, fp.k);
        bool v = false;
        if (ok) {
            n_ok++;
            if (reuse && ((sure[p >> 6] >> (p & 63)) & 1ULL)) {
                v = true;
                n_reused++;
            } else {
                v = fd_bloom_contains_canon(bloom, fd_canon(fd_kmer_at(codes, p, fp.k), fp.k), fp.tai_mask, fp.n_hash);
            }
        }
        uint64_t vm = __ballot(v);
        if (fd_lane() == 0) valid[p >> 6] = vm;
    }
    block_add(&cnt->kmers, n_ok);
    block_add(&cnt->valid_reused, n_reused);
}

// One thread per 64-position word of the valid plane: run starts by bit arithmetic; each start measures its run and,
// if it is a piece (>= k windows, inside a long enough segment), publishes ps/pm.
__global__ void __launch_bounds__(256) k_scan_pieces(const uint64_t* __restrict__ valid, const uint64_t* __restrict__ bad,
                                                     uint64_t n_words, FdParams fp, int check_segment,
                                                     unsigned long long* pm, uint64_t* __restrict__ ps, DevCounters* cnt) {
    unsigned long long n_pieces = 0;
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t vw = valid[w];
        uint64_t prev = w ? valid[w - 1] >> 63 : 0;
        uint64_t starts = vw & ~((vw << 1) | prev);
        uint64_t psw = 0;
        while (starts) {
            const int sbit = __builtin_ctzll(starts);
            starts &= starts - 1;
            const uint64_t p = w * 64 + sbit;
            // run length: first 0 bit at or after p (zero padding past the end terminates the scan)
            uint64_t q = p, len = 0;
            for (;;) {
                uint64_t bits = ~fd_bits_at(valid, q);
                if (bits) { len += __builtin_ctzll(bits); break; }
                len += 64;
                q += 64;
            }
            bool piece = len >= (uint64_t)fp.k;
            if (piece && check_segment) {
                // the unambiguous segment around the run must have length >= k + 2j + 1 (ReadScanner.cpp:268)
                uint64_t b = p;            // walk back to the segment start
                while (b > 0 && !((bad[(b - 1) >> 6] >> ((b - 1) & 63)) & 1ULL)) b--;
                uint64_t e = p;            // forward to the first bad position
                for (;;) {
                    uint64_t bits = fd_bits_at(bad, e);
                    if (bits) { e += __builtin_ctzll(bits); break; }
                    e += 64;
                }
                piece = (e - b) >= (uint64_t)(fp.k + 2 * fp.j + 1);
            }
            if (piece) {
                psw |= 1ULL << sbit;
                n_pieces++;
                // pm bits p .. p+len-1 (a word can be shared with a neighbouring piece: atomicOr)
                uint64_t a = p, z = p + len;
                while (a < z) {
                    uint64_t wi = a >> 6;
                    uint64_t hi = (wi + 1) << 6;
                    uint64_t upto = z < hi ? z : hi;
                    int lo_b = (int)(a & 63), n_b = (int)(upto - a);
                    unsigned long long m = (n_b == 64 ? ~0ULL : ((1ULL << n_b) - 1)) << lo_b;
                    atomicOr(&pm[wi], m);
                    a = upto;
                }
            }
        }
        ps[w] = psw;
    }
    block_add(&cnt->pieces, n_pieces);
}

// exclusive prefix sum of popcount(ps[w]) over the words of the batch: three small kernels
// (per-block sums, scan of the block sums by one block, apply).
constexpr int SCAN_BLOCK = 1024;
__global__ void __launch_bounds__(SCAN_BLOCK) k_prefix_block(const uint64_t* __restrict__ ps, uint64_t n_words,
                                                             uint32_t* __restrict__ prefix, uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t sh[SCAN_BLOCK];
    uint64_t w = (uint64_t)blockIdx.x * SCAN_BLOCK + threadIdx.x;
    uint32_t c = w < n_words ? (uint32_t)__popcll(ps[w]) : 0;
    sh[threadIdx.x] = c;
    __syncthreads();
    for (int o = 1; o < SCAN_BLOCK; o <<= 1) {
        uint32_t t = threadIdx.x >= (unsigned)o ? sh[threadIdx.x - o] : 0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    if (w < n_words) prefix[w] = sh[threadIdx.x] - c;   // exclusive within the block
    if (threadIdx.x == SCAN_BLOCK - 1) block_sums[blockIdx.x] = sh[threadIdx.x];
}

__global__ void __launch_bounds__(SCAN_BLOCK) k_prefix_sums(uint32_t* block_sums, uint32_t n_blocks) {
    // serial-over-chunks scan by one block; n_blocks is small (n_words / 1024)
    __shared__ uint32_t sh[SCAN_BLOCK];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n_blocks; base += SCAN_BLOCK) {
        uint32_t i = base + threadIdx.x;
        uint32_t c = i < n_blocks ? block_sums[i] : 0;
        sh[threadIdx.x] = c;
        __syncthreads();
        for (int o = 1; o < SCAN_BLOCK; o <<= 1) {
            uint32_t t = threadIdx.x >= (unsigned)o ? sh[threadIdx.x - o] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < n_blocks) block_sums[i] = carry + sh[threadIdx.x] - c;
        __syncthreads();
        if (threadIdx.x == SCAN_BLOCK - 1) carry += sh[threadIdx.x];
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) k_prefix_apply(uint32_t* __restrict__ prefix, const uint32_t* __restrict__ block_sums,
                                                      uint64_t n_words) {
    uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w < n_words) prefix[w] += block_sums[w / SCAN_BLOCK];
}

// piece list: thread on a piece start writes {start, windows} at its rank
__global__ void __launch_bounds__(256) k_scan_piece_list(const uint64_t* __restrict__ ps, const uint64_t* __restrict__ pm,
                                                         const uint32_t* __restrict__ prefix, uint64_t n_words, uint2* __restrict__ pieces) {
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t word = ps[w];
        uint32_t rank = prefix[w];
        while (word) {
            const int sbit = __builtin_ctzll(word);
            word &= word - 1;
            const uint64_t p = w * 64 + sbit;
            uint64_t q = p, len = 0;
            for (;;) {   // run of pm bits; pieces are separated by at least one 0
                uint64_t bits = ~fd_bits_at(pm, q);
                if (bits) { len += __builtin_ctzll(bits); break; }
                len += 64;
                q += 64;
            }
            pieces[rank++] = make_uint2((uint32_t)p, (uint32_t)len);
        }
    }
}

// read (index in the batch) every piece lies on: largest i with S_i <= piece start, S_i = offs[i] - offs[0] + i
__global__ void __launch_bounds__(256) k_scan_piece_read(const uint2* __restrict__ pieces, uint64_t n_pieces, const uint64_t* __restrict__ offs,
                                                         uint64_t n_reads, uint32_t* __restrict__ piece_read) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pieces) return;
    const uint64_t off0 = offs[0], s0 = pieces[i].x;
    uint64_t lo = 0, hi = n_reads - 1;
    while (lo < hi) {
        const uint64_t mid = (lo + hi + 1) >> 1;
        if ((offs[mid] - off0) + mid <= s0) lo = mid; else hi = mid - 1;
    }
    piece_read[i] = (uint32_t)lo;
}

// testForJunction where the walk may need it.  Work item = (position, direction) with the need bit set and a neighbour
// window on that side.  Only a third to a half of the positions qualify, so a lane-per-position layout would leave most
// lanes of a wave idle while the others wait for their probes: instead a wave takes 256 positions at a time, gathers the
// two work masks of each of the four words, and hands the set bits out densely, 64 items per round (rank -> word by
// prefix counts, rank within the word -> bit by popcount bisection).  Results are rare non-zero bits: atomicOr into
// zero-initialised planes.
__global__ void __launch_bounds__(256) k_scan_flags(const uint64_t* __restrict__ codes, const uint64_t* __restrict__ pm,
                                                    const uint64_t* __restrict__ need, uint64_t T, uint64_t n_words, FdParams fp, const uint32_t* __restrict__ bloom,
                                                    unsigned long long* ff, unsigned long long* fb, unsigned long long* cf0,
                                                    unsigned long long* cf1, unsigned long long* cb0, unsigned long long* cb1,
                                                    DevCounters* cnt) {
    const int lane = fd_lane();
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint64_t n_chunks = (n_words + 3) / 4;
    unsigned long long n_eval = 0, n_piece = 0;
    for (uint64_t chunk = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; chunk < n_chunks; chunk += n_waves) {
        // lanes 0..3: forward work mask of word 4*chunk + lane; lanes 4..7: backward work mask of word 4*chunk + lane - 4
        uint64_t mine = 0;
        if (lane < 8) {
            const uint64_t w = chunk * 4 + (lane & 3);
            if (w < n_words) {
                const uint64_t tmask = (w + 1) * 64 <= T ? ~0ULL : (T > w * 64 ? (1ULL << (T - w * 64)) - 1 : 0ULL);
                const uint64_t pmw = pm[w] & tmask, nd = need[w] & tmask;
                if (lane < 4) {
                    mine = nd & pmw & ((pmw >> 1) | (pm[w + 1] << 63));          // a window follows: facing forward
                    n_eval += __popcll(nd);
                    n_piece += __popcll(pmw);
                } else {
                    mine = nd & pmw & ((pmw << 1) | (w ? pm[w - 1] >> 63 : 0ULL));   // a window precedes: facing backward
                }
            }
        }
        uint64_t m[8];
        int cum[9];
        cum[0] = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            m[q] = __shfl(mine, q, 64);
            cum[q + 1] = cum[q] + __popcll(m[q]);
        }
        for (int t = lane; t < cum[8]; t += 64) {
            // which mask holds item t (uniform bounds, per-lane t), which bit of it
            int q = 0;
            uint64_t mq = m[0];
            int before = 0;
#pragma unroll
            for (int c = 1; c < 8; c++)
                if (t >= cum[c]) { q = c; mq = m[c]; before = cum[c]; }
            const int bit = select_bit(mq, t - before);
            const uint64_t w = chunk * 4 + (q & 3);
            const uint64_t pos = w * 64 + bit;
            const uint64_t km = fd_kmer_at(codes, pos, fp.k);
            bool flag;
            int njc;
            const unsigned long long bm = 1ULL << bit;
            if (q < 4) {   // facing forward: real extension = base after the window (utils/ReadKmer.cpp:107-110)
                test_for_junction(km, fd_base_at(codes, pos + fp.k), fp, bloom, flag, njc);
                if (flag) atomicOr(&ff[w], bm);
                if (njc & 1) atomicOr(&cf0[w], bm);
                if (njc & 2) atomicOr(&cf1[w], bm);
            } else {       // facing backward: reverse complement, real extension = complement of the base before (:111-113)
                test_for_junction(fd_revcomp(km, fp.k), fd_base_at(codes, pos - 1) ^ 2, fp, bloom, flag, njc);
                if (flag) atomicOr(&fb[w], bm);
                if (njc & 1) atomicOr(&cb0[w], bm);
                if (njc & 2) atomicOr(&cb1[w], bm);
            }
        }
    }
    block_add(&cnt->flag_positions, n_eval);
    block_add(&cnt->piece_positions, n_piece);
}

// The same work as k_scan_flags for j <= 1, organised as a per-lane state machine.  An item is a chain of 3 to ~20 DEPENDENT bit
// tests with early exits (3 alternate extensions, each up to n_hash bits, each present one followed by up to 4 x n_hash bits of
// jcheck), so in the kernel above a wave runs as long as its longest chain while most of its lanes have long finished: the
// probes in flight -- the only thing a random-access-bound kernel has -- drop to under half of the lanes.  Here every lane
// issues exactly one bit test per iteration and a lane whose chain has ended takes the next item in the same iteration.  Items
// come from a pool the wave shares: 64 words (one per lane) of forward / backward work masks in LDS with their prefix counts;
// item t is found by bisection over the prefix counts.  The pool rolls on to the next 64 words as soon as it is empty, while
// unfinished chains of the previous words are still running (they hold everything they need in registers).
struct FlagPool {
    unsigned long long mf[64], mb[64];
    int excl[64];
};

// chain state of one work item (8 registers): S of them per lane, so that a lane has S independent bit tests in flight
struct FlagChain {
    uint64_t key;        // the k-mer, oriented towards the extension under test
    uint64_t hA, hB;     // hashes of the k-mer being looked up (the alternate, or one of its extensions during jcheck)
    uint32_t w;          // word of the batch the item sits in
    uint32_t st;         // packed: bit 0 active, 1 fresh, 2 backward, 3-4 real, 5-7 nt, 8-10 jnt+1 (0 = testing the alternate itself),
                         //         11-14 h, 15-16 njc, 17-22 bit
};
#define FC_ACTIVE 1u
#define FC_FRESH 2u
#define FC_BACKWARD 4u
#define FC_GET(st, sh, bits) (((st) >> (sh)) & ((1u << (bits)) - 1))
#define FC_SET(st, sh, bits, v) ((st) = ((st) & ~(((1u << (bits)) - 1) << (sh))) | ((uint32_t)(v) << (sh)))

template <int S>
__global__ void __launch_bounds__(256) k_scan_flags_sm(const uint64_t* __restrict__ codes, const uint64_t* __restrict__ pm,
                                                       const uint64_t* __restrict__ need, uint64_t T, uint64_t n_words, FdParams fp,
                                                       const uint32_t* __restrict__ bloom, unsigned long long* ff, unsigned long long* fb,
                                                       unsigned long long* cf0, unsigned long long* cf1, unsigned long long* cb0,
                                                       unsigned long long* cb1, DevCounters* cnt) {
    // Plain LDS objects indexed directly, so that the accesses are ds_read / ds_write: those execute in order for a wave, which is
    // what makes a word written by one lane visible to the lane that reads it next.  (Declared volatile, or reached through a
    // generic pointer, they become FLAT accesses, whose order between the lanes of a wave is NOT guaranteed.)  The wavefront-scope
    // fences around the refill keep the compiler from moving or caching the accesses across it.
    __shared__ FlagPool pools[4];
    const int wid = (int)(threadIdx.x >> 6);
    const int lane = fd_lane();
    const uint64_t lt_mask = (1ULL << lane) - 1;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint64_t wv = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_groups = (n_words + 63) / 64;
    uint64_t grp = wv * n_groups / n_waves;
    const uint64_t grp_end = (wv + 1) * n_groups / n_waves;
    unsigned long long n_eval = 0, n_piece = 0;
    int next = 0, total = 0;          // uniform: items of the pool handed out / in the pool
    uint64_t pool_word0 = 0;          // uniform: first word of the pool
    FlagChain ch[S];
#pragma unroll
    for (int q = 0; q < S; q++) ch[q].st = 0;

    for (;;) {
        // ---- hand items to the slots without one
        uint64_t idle[S];
        uint64_t any_idle = 0;
#pragma unroll
        for (int q = 0; q < S; q++) { idle[q] = __ballot(!(ch[q].st & FC_ACTIVE)); any_idle |= idle[q]; }
        while (any_idle && (next < total || grp < grp_end)) {
            if (next == total) {   // the pool is empty: the next 64 words
                const uint64_t w = grp * 64 + lane;
                uint64_t f = 0, b = 0;
                if (w < n_words) {
                    const uint64_t tmask = (w + 1) * 64 <= T ? ~0ULL : (T > w * 64 ? (1ULL << (T - w * 64)) - 1 : 0ULL);
                    const uint64_t pmw = pm[w] & tmask, nd = need[w] & tmask;
                    f = nd & pmw & ((pmw >> 1) | (pm[w + 1] << 63));              // a window follows: facing forward
                    b = nd & pmw & ((pmw << 1) | (w ? pm[w - 1] >> 63 : 0ULL));   // a window precedes: facing backward
                    n_eval += __popcll(nd);
                    n_piece += __popcll(pmw);
                }
                const int c = __popcll(f) + __popcll(b);
                int incl = c;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int o = __shfl_up(incl, d, 64);
                    if (lane >= d) incl += o;
                }
                pools[wid].mf[lane] = f;
                pools[wid].mb[lane] = b;
                pools[wid].excl[lane] = incl - c;
                total = __builtin_amdgcn_readlane(incl, 63);
                next = 0;
                pool_word0 = grp * 64;
                grp++;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                continue;
            }
#pragma unroll
            for (int q = 0; q < S; q++) {   // slot 0 of every lane first, then slot 1 ...: item t goes to the t-th idle slot in that order
                const int r = __popcll(idle[q] & lt_mask);
                const int avail = total - next;
                if (!(ch[q].st & FC_ACTIVE) && r < avail) {
                    const int t = next + r;
                    int sl = 0;
#pragma unroll
                    for (int step = 32; step > 0; step >>= 1)
                        if (pools[wid].excl[sl + step] <= t) sl += step;   // largest sl with excl[sl] <= t (words without items share the excl of the next)
                    const uint64_t f = pools[wid].mf[sl], b = pools[wid].mb[sl];
                    const int within = t - pools[wid].excl[sl], cf = __popcll(f);
                    const bool backward = within >= cf;
                    const int bit = backward ? select_bit(b, within - cf) : select_bit(f, within);
                    const uint64_t iw = pool_word0 + sl;
                    const uint64_t pos = iw * 64 + bit;
                    const uint64_t km = fd_kmer_at(codes, pos, fp.k);
                    int real;
                    if (!backward) {   // real extension = base after the window (utils/ReadKmer.cpp:107-110)
                        ch[q].key = km;
                        real = fd_base_at(codes, pos + fp.k);
                    } else {           // reverse complement, real extension = complement of the base before (:111-113)
                        ch[q].key = fd_revcomp(km, fp.k);
                        real = fd_base_at(codes, pos - 1) ^ 2;
                    }
                    ch[q].w = (uint32_t)iw;
                    uint32_t st = FC_ACTIVE | FC_FRESH | (backward ? FC_BACKWARD : 0u);
                    FC_SET(st, 3, 2, real);
                    FC_SET(st, 5, 3, real == 0 ? 1 : 0);
                    FC_SET(st, 17, 6, bit);
                    ch[q].st = st;
                }
                next += min(__popcll(idle[q]), avail > 0 ? avail : 0);
            }
            any_idle = 0;
#pragma unroll
            for (int q = 0; q < S; q++) { idle[q] = __ballot(!(ch[q].st & FC_ACTIVE)); any_idle |= idle[q]; }
        }
        bool mine = false;
#pragma unroll
        for (int q = 0; q < S; q++) mine |= (ch[q].st & FC_ACTIVE) != 0;
        if (!__ballot(mine)) break;
        // ---- one bit test per active slot: all S loads are issued before any result is looked at
        uint32_t word[S];
        uint32_t sh[S];
#pragma unroll
        for (int q = 0; q < S; q++) {
            word[q] = 0;
            sh[q] = 0;
            if (ch[q].st & FC_ACTIVE) {
                if (ch[q].st & FC_FRESH) {   // a new k-mer to look up: the alternate itself (jnt+1 == 0) or one of its extensions (jcheck)
                    const uint32_t jn = FC_GET(ch[q].st, 8, 3);
                    const uint64_t alt = ((ch[q].key << 2) | (uint64_t)FC_GET(ch[q].st, 5, 3)) & fp.kmask;
                    const uint64_t e = jn == 0 ? alt : ((alt << 2) | (uint64_t)(jn - 1)) & fp.kmask;
                    fd_hash_pair(fd_canon(e, fp.k), fp.tai_mask, ch[q].hA, ch[q].hB);
                    FC_SET(ch[q].st, 11, 4, 0);
                    ch[q].st &= ~FC_FRESH;
                }
                const uint64_t p = (ch[q].hA + (uint64_t)FC_GET(ch[q].st, 11, 4) * ch[q].hB) & fp.tai_mask;
                word[q] = bloom[p >> 5];
                sh[q] = (uint32_t)(p & 31);
            }
        }
#pragma unroll
        for (int q = 0; q < S; q++) {
            uint32_t st = ch[q].st;
            if (st & FC_ACTIVE) {
            const bool bit = (word[q] >> sh[q]) & 1u;
            const uint32_t real = FC_GET(st, 3, 2);
            uint32_t jn = FC_GET(st, 8, 3), h = FC_GET(st, 11, 4), njc = FC_GET(st, 15, 2), nt = FC_GET(st, 5, 3);
            bool done = false, flag = false;
            if (bit && h + 1 < (uint32_t)fp.n_hash) {
                h++;                                    // next bit of the same k-mer
            } else if (bit) {                           // the k-mer is in the filter
                if (jn == 0) {                          // an alternate extension exists (src/ReadScanner.cpp:44-49)
                    njc++;
                    if (fp.j == 0) { done = true; flag = true; }
                    else { jn = 1; st |= FC_FRESH; }
                } else {                                // and it continues: junction
                    done = true;
                    flag = true;
                }
            } else {                                    // absent
                if (jn >= 1 && jn < 4) {
                    jn++;
                    st |= FC_FRESH;
                } else {                                // alternate absent, or none of its 4 extensions present: the next alternate
                    nt++;
                    if (nt == real) nt++;
                    if (nt > 3) done = true;
                    else { jn = 0; st |= FC_FRESH; }
                }
            }
            FC_SET(st, 8, 3, jn);
            FC_SET(st, 11, 4, h);
            FC_SET(st, 15, 2, njc);
            FC_SET(st, 5, 3, nt & 7);
            if (done) {
                const unsigned long long bm = 1ULL << FC_GET(st, 17, 6);
                const uint32_t iw = ch[q].w;
                if (!(st & FC_BACKWARD)) {
                    if (flag) atomicOr(&ff[iw], bm);
                    if (njc & 1) atomicOr(&cf0[iw], bm);
                    if (njc & 2) atomicOr(&cf1[iw], bm);
                } else {
                    if (flag) atomicOr(&fb[iw], bm);
                    if (njc & 1) atomicOr(&cb0[iw], bm);
                    if (njc & 2) atomicOr(&cb1[iw], bm);
                }
                st &= ~FC_ACTIVE;
            }
            ch[q].st = st;
            }
        }
    }
    block_add(&cnt->flag_positions, n_eval);
    block_add(&cnt->piece_positions, n_piece);
}

// ---- Stage 3's Bloom probes, batched: pure functions of bloo2, one k-mer per lane ----------------
// mode 0  JChecker::jcheck(kmer_type)             utils/JChecker.cpp:51-80
// mode 1  JunctionMap::getValidJExtension         utils/JunctionMap.cpp:474-490   (-1 none, -2 several, else the nucleotide)
// mode 2  JunctionMap::isBloomJunction            utils/JunctionMap.cpp:494-504
__global__ void __launch_bounds__(256) k_probe_stage3(const uint64_t* __restrict__ kmers, uint64_t n, int mode, FdParams fp,
                                                      const uint32_t* __restrict__ bloom, signed char* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t km = kmers[i] & fp.kmask;
    int r;
    if (mode == 0) {
        r = jcheck_dfs(km, fp, bloom) ? 1 : 0;
    } else if (mode == 1) {
        r = -1;
        for (int nt = 0; nt < 4; nt++) {
            const uint64_t e = ((km << 2) | (uint64_t)nt) & fp.kmask;
            if (fd_bloom_contains_canon(bloom, fd_canon(e, fp.k), fp.tai_mask, fp.n_hash) && jcheck_dfs(e, fp, bloom)) {
                if (r != -1) { r = -2; break; }
                r = nt;
            }
        }
    } else {
        int paths = 0;
        for (int nt = 0; nt < 4; nt++) paths += jcheck_dfs(((km << 2) | (uint64_t)nt) & fp.kmask, fp, bloom) ? 1 : 0;
        r = paths > 1 ? 1 : 0;
    }
    out[i] = (signed char)r;
}

}  // namespace

// fgpu_create touches one kernel of every translation unit from a helper thread: the runtime loads a unit's code object at the first use of
// one of its kernels (20-25 ms for the large units), which otherwise lands on the first batch of each pass
void fgpu_touch_scan_pure() {
    hipFuncAttributes attr;
    (void)hipFuncGetAttributes(&attr, (const void*)k_scan_same);
}

int fgpu_util_probe_stage3(fgpu_ctx* ctx, const uint64_t* d_kmers, uint64_t n, int mode, signed char* d_out) {
    FGPU_LAUNCH("probe_stage3", k_probe_stage3, fgpu_blocks(n, 256), 256, d_kmers, n, mode, ctx->fd, (const uint32_t*)ctx->bloo2, d_out);
    return FGPU_OK;
}

int fgpu_stage_scan_pure(fgpu_ctx* ctx, uint64_t* n_pieces) {
    BatchBufs& bb = *ctx->cur;
    *n_pieces = 0;
    bb.n_pieces = 0;   // the buffers are recycled: an empty batch must not inherit the previous batch's pieces
    if (bb.T == 0) return FGPU_OK;
    const uint64_t wb = (bb.n_words + FGPU_PADW) * 8;
    int rc;
    DevBuf* planes[] = {&bb.valid, &bb.pm, &bb.ps, &bb.ff, &bb.fb, &bb.cf0, &bb.cf1, &bb.cb0, &bb.cb1, &bb.inF, &bb.inB, &bb.lk, &bb.nF, &bb.nB, &bb.need};
    for (DevBuf* b : planes)
        if ((rc = fgpu_ensure_b(ctx, b, wb))) return rc;
    if ((rc = fgpu_ensure_b(ctx, &bb.ps_prefix, (bb.n_words + FGPU_PADW + bb.n_words / SCAN_BLOCK + 2) * 4))) return rc;
    const double t_a = fgpu_host_now();
    // valid/pm/ps need zeroed padding (funnel reads run one word past the end); pm is built with atomicOr
    FGPU_HIP(hipMemsetAsync(bb.valid.p, 0, wb, ctx->stream));
    FGPU_HIP(hipMemsetAsync(bb.pm.p, 0, wb, ctx->stream));
    FGPU_HIP(hipMemsetAsync(bb.ps.p, 0, wb, ctx->stream));

    // kernels with one lane per position (capping the grid to leave wave slots for the concurrently running walk of
    // the previous batch was measured and does not pay: the walk is slowed by memory contention, not by slots)
    const unsigned grid = fgpu_grid(bb.n_words * 64, 256);
    const unsigned wgrid = fgpu_grid(bb.n_words, 256);            // kernels with one thread per 64-position word
    if ((rc = fgpu_util_count_segments(ctx, ctx->fd.k + 2 * ctx->fd.j + 1))) return rc;
    // the load pass' batch of the same index, if it was kept and has the same shape: compare the streams on the device
    const ResidentBatch* kept = ctx->scan_batch_index < ctx->resident_count ? ctx->resident[ctx->scan_batch_index] : nullptr;
    ctx->scan_batch_index++;
    if (kept && (kept->T != bb.T || kept->n_words != bb.n_words)) kept = nullptr;
    if (kept) {
        if ((rc = fgpu_ensure_b(ctx, &bb.same, 64))) return rc;
        FGPU_HIP(hipMemsetAsync(bb.same.p, 0x01, 4, ctx->stream));
        const uint64_t ncw = 2 * bb.n_words;   // 32 bases per code word
        FGPU_LAUNCH("scan_same", k_scan_same, fgpu_grid(ncw, 256), 256, (const uint64_t*)bb.codes.p, (const uint64_t*)kept->codes.p, ncw,
                    (const uint64_t*)bb.bad.p, (const uint64_t*)kept->bad.p, bb.n_words, (uint32_t*)bb.same.p);
    }
    FGPU_LAUNCH("scan_valid", k_scan_valid, grid, 256, (const uint64_t*)bb.codes.p, (const uint64_t*)bb.bad.p, bb.T, bb.n_words, ctx->fd,
                (const uint32_t*)ctx->bloo2, kept ? (const uint64_t*)kept->sure.p : (const uint64_t*)nullptr,
                kept ? (const uint32_t*)bb.same.p : (const uint32_t*)nullptr, (uint64_t*)bb.valid.p, ctx->counters);
    const int check_segment = ctx->fd.k < 2 * ctx->fd.j + 2;   // otherwise a run of k windows already implies the length gate
    FGPU_LAUNCH("scan_pieces", k_scan_pieces, wgrid, 256, (const uint64_t*)bb.valid.p, (const uint64_t*)bb.bad.p, bb.n_words, ctx->fd,
                check_segment, (unsigned long long*)bb.pm.p, (uint64_t*)bb.ps.p, ctx->counters);
    uint32_t* prefix = (uint32_t*)bb.ps_prefix.p;
    uint32_t* block_sums = prefix + bb.n_words + FGPU_PADW;
    const uint32_t nblk = fgpu_blocks(bb.n_words + 1, SCAN_BLOCK);
    // one word more than the batch has, so that prefix[n_words] = number of pieces (rank queries at position T)
    const uint64_t nw1 = bb.n_words + 1;
    FGPU_LAUNCH("prefix_block", k_prefix_block, nblk, SCAN_BLOCK, (const uint64_t*)bb.ps.p, nw1, prefix, block_sums);
    FGPU_LAUNCH("prefix_sums", k_prefix_sums, 1, SCAN_BLOCK, block_sums, nblk);
    FGPU_LAUNCH("prefix_apply", k_prefix_apply, fgpu_blocks(nw1, 256), 256, prefix, (const uint32_t*)block_sums, nw1);
    FGPU_HIP(hipMemcpyAsync(ctx->counters_host, ctx->counters, sizeof(DevCounters), hipMemcpyDeviceToHost, ctx->stream));
    const double t_b = fgpu_host_now();
    FGPU_HIP(fgpu_sync_stream(ctx, ctx->stream));
    const double t_c = fgpu_host_now();
    ctx->host_ms[2] += t_b - t_a;
    ctx->host_ms[3] += t_c - t_b;
    bb.max_piece_span = ctx->counters_host->max_read_len + 64;
    const uint64_t np = ctx->counters_host->pieces - ctx->scan_pieces_seen;
    ctx->scan_pieces_seen = ctx->counters_host->pieces;
    *n_pieces = np;
    bb.n_pieces = np;
    if ((rc = fgpu_ensure_b(ctx, &bb.pieces, (np + 1) * sizeof(uint2)))) return rc;
    if (np) {
        FGPU_LAUNCH("piece_list", k_scan_piece_list, wgrid, 256, (const uint64_t*)bb.ps.p, (const uint64_t*)bb.pm.p,
                    (const uint32_t*)prefix, bb.n_words, (uint2*)bb.pieces.p);
        if (ctx->record_stops) {   // the offsets are the caller's and only valid during this call
            if ((rc = fgpu_ensure_b(ctx, &bb.piece_read, np * 4))) return rc;
            FGPU_LAUNCH("piece_read", k_scan_piece_read, fgpu_blocks(np, 256), 256, (const uint2*)bb.pieces.p, np, bb.d_offs, bb.n_reads,
                        (uint32_t*)bb.piece_read.p);
        }
        if ((rc = fgpu_stage_scan_need(ctx))) return rc;
        DevBuf* outs[] = {&bb.ff, &bb.fb, &bb.cf0, &bb.cf1, &bb.cb0, &bb.cb1};
        for (DevBuf* o : outs) FGPU_HIP(hipMemsetAsync(o->p, 0, wb, ctx->stream));
        // twice the usual grid: shorter-lived blocks free wave slots more often, which lets the high-priority walk kernels of the
        // previous batch in sooner (walk stage 86 -> 80 ms per step; beyond 16 K blocks the flags kernel itself slows down)
        const unsigned flags_grid = (unsigned)std::min<uint64_t>(std::max<uint64_t>(bb.n_words / 16, 1), 2 * FGPU_GRID_BLOCKS);
        static const int sm_blocks = getenv("FGPU_FLAGS_SM_BLOCKS") ? atoi(getenv("FGPU_FLAGS_SM_BLOCKS")) : 4096;
        if (ctx->fd.j <= 1 && sm_blocks > 0) {
            const unsigned sm_grid = (unsigned)std::min<uint64_t>(std::max<uint64_t>((bb.n_words + 255) / 256, 1), (uint64_t)sm_blocks);
            static const int sm_slots = getenv("FGPU_FLAGS_SM_SLOTS") ? atoi(getenv("FGPU_FLAGS_SM_SLOTS")) : 1;
#define FGPU_FLAGS_SM(SLOTS)                                                                                                        \
    FGPU_LAUNCH("scan_flags", k_scan_flags_sm<SLOTS>, sm_grid, 256, (const uint64_t*)bb.codes.p, (const uint64_t*)bb.pm.p,           \
                (const uint64_t*)bb.need.p, bb.T, bb.n_words, ctx->fd, (const uint32_t*)ctx->bloo2, (unsigned long long*)bb.ff.p,    \
                (unsigned long long*)bb.fb.p, (unsigned long long*)bb.cf0.p, (unsigned long long*)bb.cf1.p,                         \
                (unsigned long long*)bb.cb0.p, (unsigned long long*)bb.cb1.p, ctx->counters)
            if (sm_slots <= 1) FGPU_FLAGS_SM(1);
            else if (sm_slots == 2) FGPU_FLAGS_SM(2);
            else if (sm_slots == 3) FGPU_FLAGS_SM(3);
            else FGPU_FLAGS_SM(4);
#undef FGPU_FLAGS_SM
        } else
        FGPU_LAUNCH("scan_flags", k_scan_flags, flags_grid, 256, (const uint64_t*)bb.codes.p, (const uint64_t*)bb.pm.p,
                    (const uint64_t*)bb.need.p, bb.T, bb.n_words, ctx->fd, (const uint32_t*)ctx->bloo2, (unsigned long long*)bb.ff.p,
                    (unsigned long long*)bb.fb.p, (unsigned long long*)bb.cf0.p, (unsigned long long*)bb.cf1.p,
                    (unsigned long long*)bb.cb0.p, (unsigned long long*)bb.cb1.p, ctx->counters);
        if ((rc = fgpu_stage_scan_debug_drop(ctx))) return rc;
    }
    ctx->host_ms[4] += fgpu_host_now() - t_c;
    return FGPU_OK;
}
