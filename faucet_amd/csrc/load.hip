// load.hip — pass 1: the two-filter Bloom load, exact and order-free.
//
// Replaces the loop body of load_two_filters (utils/Bloom.cpp:289-299):
//     for every k-mer window, in processing order:
//         if (bloo1->contains(hA,hB)) bloo2->add(hA,hB); else bloo1->add(hA,hB);
//
// Exact parallel form.  Because an occurrence either finds all of its bits set or
// sets them all, bloo1 before time t is the OR of the bits of all occurrences < t.  Hence
//     occurrence t goes to bloo2  <=>  every one of its bits was first set at a time < t.
// Per batch:
//   k_load_mark   (all windows)     test the carried-in bitmap (bloo1 as it stood before this batch).
//                                   All bits set  -> the occurrence certainly goes to bloo2: set its bits there.
//                                   Otherwise     -> atomicMin(first[bit], t) for the bits not yet in the carry,
//                                                    flag the occurrence as pending.
//   k_load_resolve (pending only)   bit is "set before t" iff it is in the carry or first[bit] < t;
//                                   all bits set before t -> bloo2.
//   k_carry_from_first              carry |= bits whose first-set time is no longer "never" (one sweep of first[]) -- not after
//                                   every batch: the carry may lag (fgpu_stage_load), a sweep closes an EPOCH of batches.
// Both kernels also write the `sure` plane (occurrence routed to bloo2), which the scan of the same reads reuses.
// t is the stream position within the epoch (pack.hip makes position order == processing order); first[] never needs a
// reset because a bit that is still 0 in the carry has first[bit] == 0xFFFFFFFF or a time of the current epoch.
// Multi-GPU shards may ask for times that count through the whole pass instead (FGPU_LOAD_SHARD_TIMES) and re-evaluate
// afterwards what they kept out of bloo2 against the lower ranks' bits (k_load_fixup).
//
// Launch shape: a fixed grid of FGPU_GRID_BLOCKS x 256 threads strides over the stream, lanes = consecutive
// positions (so every per-position plane is written as one 8-byte ballot word per wave) and counters are
// kept in registers until the wave retires: one atomic per wave per kernel instead of one per 64 positions
// (same-address atomics serialise at ~10 ns each; 16 M of them per 10^9 positions cost more than the probes).
//
// Roofline: HBM/Infinity-Cache random access.  Algorithmic bytes per k-mer:
//   L/(L-k+1) bytes of bases + 64 B * n_hash (test-and-set on bloo1) + 64 B * rho * n_hash (set on bloo2).
#include <algorithm>

#include "fgpu_ctx.h"
#include "fgpu_flags.h"

namespace {

__device__ __forceinline__ void wave_add(unsigned long long* dst, unsigned long long v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (fd_lane() == 0 && v) atomicAdd(dst, v);
}

// one atomic per BLOCK: the waves of a grid-stride kernel retire together, and 16 K of them adding to one word is a queue of
// ~10 ns same-address atomics at the tail of every launch (k_scan_pieces spent most of its time in it).  All threads must call.
__device__ __forceinline__ void block_add(unsigned long long* dst, unsigned long long v) {
    __shared__ unsigned long long part[4];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (fd_lane() == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t = part[0] + part[1] + part[2] + part[3];
        if (t) atomicAdd(dst, t);
    }
    __syncthreads();   // part[] is reused by the next call
}

// During a load pass the two filters live INTERLEAVED: pair[w] = {word w of the carried-in bloo1, word w of bloo2}.
// Both filters use the same bit positions (same hashes, same size), so one 8-byte load serves the carry test and
// the test-before-set of bloo2: 3 random loads per k-mer instead of up to 6.  fgpu_load_end splits them again.
// -DFGPU_FIRST_MASK_LOG2=n (MEASUREMENT build, results wrong): first-set times folded into a table of 2^n entries -- what a first[] of that
// size would cost k_load_mark's atomics and k_load_resolve's loads, without building the structure that would make it exact
#ifdef FGPU_FIRST_MASK_LOG2
#define FD_FIRST(h) ((h) & ((1ULL << FGPU_FIRST_MASK_LOG2) - 1))
#else
#define FD_FIRST(h) (h)
#endif
constexpr int MISS_PLANES = 4;   // planes of "bit i was missing from the carry" kept for k_load_resolve (hash functions beyond are re-tested)

// Where the working state of a load pass lives.  Two layouts, one algorithm:
//   REC = 0  `base` = pair[]: {bloo1 word, bloo2 word} interleaved, 8 bytes per 32 filter bits; the first-set times in their own array first[]
//            (4 bytes per filter bit).  Filters up to 2^31 bits: the 8-byte words of config 2 (128 MiB) stay in the Infinity Cache.
//   REC = 1  `base` = 256-byte RECORDS, one per 32 filter bits, aligned: word 0 bloo1, word 1 bloo2, words 16..47 the first-set times of the
//            record's 32 bits (two further lines of the same 256 bytes), the rest unused: FGPU_LOAD_LAYOUT=records (round 5; measured, NOT the
//            default).  The marking kernel is bound by the atomicMin of the times it posts (1.9 per k-mer on configs 4 and 5), and an atomic
//            into the 256-byte block whose first line the kernel has just loaded costs half of one into a separate 32 GiB array -- same three
//            loads, same three atomics per new k-mer, other addresses (scripts/micro/mark_model3.hip: 2^33 bits, 63 % new k-mers 21.8 -> 15.8 ms
//            per 1.34e8 k-mers; blocks of 192 or 160 bytes 17.6 / 18.5: the alignment counts; 2^29 bits 7.6 -> 13.0 ms).  On config 4's reads
//            the kernel gains 14 % (231 -> 203 ms per 25 M reads), config 5's 15 %, and the pass gives it back: a sweep that brings the carry up
//            to date streams the records (20.5 ms against 8.9), a pass begins by writing 48 GiB of them (29 ms) -- profiles/r05_load_layouts.txt,
//            DESIGN.md section 10.  64 GiB instead of 34 at 2^33 bits.
template <int REC>
struct Filt {
    uint32_t* base;
    uint32_t* first;
    __device__ __forceinline__ uint32_t* word(uint64_t h) const { return base + (REC ? ((h >> 5) << 6) : ((h >> 5) << 1)); }   // -> {bloo1, bloo2}
    __device__ __forceinline__ uint2 load(uint64_t h) const { return *(const uint2*)word(h); }
    __device__ __forceinline__ uint32_t* time(uint64_t h) const { return REC ? base + ((h >> 5) << 6) + 16 + (h & 31) : first + FD_FIRST(h); }
};
constexpr uint64_t REC_WORDS = 64;      // 32-bit words per record

template <int REC>
__global__ void __launch_bounds__(256) k_load_mark(const uint64_t* __restrict__ codes, const uint64_t* __restrict__ bad,
                                                   uint64_t T, uint64_t n_words, FdParams fp, Filt<REC> f, uint32_t tb,
                                                   uint64_t* __restrict__ pending, uint64_t plane_stride, uint64_t* __restrict__ sure, DevCounters* cnt) {
    unsigned long long n_ok = 0, n_hit = 0, n_pend = 0;
    const uint64_t total = n_words * 64;
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (uint64_t)gridDim.x * blockDim.x) {
        bool ok = p < T && fd_window_ok(bad, p, fp.k);
        bool pend = false, hit = false;
        uint32_t miss_bits = 0;
        if (ok) {
            n_ok++;
            uint64_t canon = fd_canon(fd_kmer_at(codes, p, fp.k), fp.k);
            uint64_t hA, hB;
            fd_hash_pair(canon, fp.tai_mask, hA, hB);
            // gather both filters' bits of all n_hash positions first (independent loads in flight)
            uint32_t missing = 0, b2_missing = 0;
            uint64_t h = hA;
            for (int i = 0; i < fp.n_hash; i++) {
                const uint2 v = f.load(h);
                if (!((v.x >> (h & 31)) & 1u)) missing |= 1u << i;
                if (!((v.y >> (h & 31)) & 1u)) b2_missing |= 1u << i;
                h = (h + hB) & fp.tai_mask;
            }
            if (!missing) {
                n_hit++;
                hit = true;
                if (b2_missing) {   // a stale 0 only costs a redundant atomic; bits are never cleared
                    h = hA;
                    for (int i = 0; i < fp.n_hash; i++) {
                        if (b2_missing & (1u << i)) atomicOr(f.word(h) + 1, 1u << (h & 31));
                        h = (h + hB) & fp.tai_mask;
                    }
                }
            } else {
                pend = true;
                n_pend++;
                miss_bits = missing;
                h = hA;
                for (int i = 0; i < fp.n_hash; i++) {
                    // the next carry is derived afterwards: from first[] by a sweep (k_carry_from_first) or by re-hashing the
                    // occurrences that were not contained (k_carry_set)
                    if (missing & (1u << i)) atomicMin(f.time(h), tb + (uint32_t)p);
                    h = (h + hB) & fp.tai_mask;
                }
            }
        }
        uint64_t pm = __ballot(pend), hm = __ballot(hit);
        // which of the first MISS_PLANES bits were missing from the carry: k_load_resolve starts from that instead of loading the
        // same words again (1.5 of its ~2.5 accesses per pending occurrence)
        uint64_t mm[MISS_PLANES];
#pragma unroll
        for (int i = 0; i < MISS_PLANES; i++) mm[i] = __ballot(pend && ((miss_bits >> i) & 1u));
        if (fd_lane() == 0) {
            pending[p >> 6] = pm;
            sure[p >> 6] = hm;
            if (pm) {
#pragma unroll
                for (int i = 0; i < MISS_PLANES; i++) pending[(i + 1) * plane_stride + (p >> 6)] = mm[i];
            }
        }
    }
    block_add(&cnt->kmers, n_ok);
    block_add(&cnt->to_bloo2, n_hit);
    block_add(&cnt->mark_hits, n_hit);
    block_add(&cnt->mark_pending, n_pend);
}

template <int REC>
__global__ void __launch_bounds__(256) k_load_resolve(const uint64_t* __restrict__ codes, uint64_t T, uint64_t n_words, FdParams fp,
                                                      Filt<REC> f, uint32_t tb,
                                                      const uint64_t* __restrict__ pending, uint64_t plane_stride, uint64_t* __restrict__ sure, DevCounters* cnt,
                                                      uint64_t* __restrict__ fail) {
    // fail != nullptr (read shards, fgpu_load_fixup): MISS_PLANES planes of "bit i of this occurrence was NOT set before it" -- every missing bit
    // is looked at then, not only up to the first that fails; needs n_hash <= MISS_PLANES
    unsigned long long n_pass = 0;
    const uint64_t total = n_words * 64;
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t pw = pending[p >> 6];   // wave-uniform: the 64 lanes of a wave cover one word
        if (!pw) {
            if (fail && fd_lane() == 0)
                for (int i = 0; i < MISS_PLANES; i++) fail[i * plane_stride + (p >> 6)] = 0;
            continue;
        }
        uint32_t failed = 0;
        bool pass = false;
        if ((pw >> (p & 63)) & 1ULL) {
            uint32_t missing = 0;                // what k_load_mark saw (the carry does not change between the two kernels)
#pragma unroll
            for (int i = 0; i < MISS_PLANES; i++)
                missing |= (uint32_t)((pending[(i + 1) * plane_stride + (p >> 6)] >> (p & 63)) & 1ULL) << i;
            uint64_t canon = fd_canon(fd_kmer_at(codes, p, fp.k), fp.k);
            uint64_t hA, hB;
            fd_hash_pair(canon, fp.tai_mask, hA, hB);
            pass = true;
            uint64_t h = hA;
            for (int i = 0; i < fp.n_hash; i++) {
                bool in_carry = i < MISS_PLANES ? !((missing >> i) & 1u) : ((f.load(h).x >> (h & 31)) & 1u) != 0;
                if (!in_carry && !(*f.time(h) < tb + (uint32_t)p)) {
                    pass = false;
                    failed |= 1u << i;
                    if (!fail) break;
                }
                h = (h + hB) & fp.tai_mask;
            }
            if (pass) {   // rare: every bit was set earlier in this very batch
                n_pass++;
                h = hA;
                for (int i = 0; i < fp.n_hash; i++) {
                    const uint32_t bit = 1u << (h & 31);
                    if (!(f.word(h)[1] & bit)) atomicOr(f.word(h) + 1, bit);   // a stale 0 only costs a redundant atomic
                    h = (h + hB) & fp.tai_mask;
                }
            }
        }
        const uint64_t sm = __ballot(pass);
        if (fd_lane() == 0 && sm) sure[p >> 6] |= sm;
        if (fail) {
            uint64_t fm[MISS_PLANES];
#pragma unroll
            for (int i = 0; i < MISS_PLANES; i++) fm[i] = __ballot((failed >> i) & 1u);
            if (fd_lane() == 0) {
#pragma unroll
                for (int i = 0; i < MISS_PLANES; i++) fail[i * plane_stride + (p >> 6)] = fm[i];
            }
        }
    }
    block_add(&cnt->to_bloo2, n_pass);
}

// The same resolution with every lane busy.  Only a quarter of the positions are pending, so in the kernel above a wave has ~17
// loads in flight instead of 64, and the loads go to first[] (4 bytes per filter bit: 2 GiB and more), the slowest table of the
// path: the kernel sat at a third of the random-access rate.  Here a wave shares a pool of 64 pending words (LDS, prefix counts,
// bisection -- the scheme of k_scan_flags_sm) and every lane carries one pending occurrence at a time: one first[] load per
// iteration for its next missing bit, and a lane that is done (almost always after the first load: a k-mer that is new in this
// batch set that time itself) takes the next occurrence in the same iteration.  Needs the missing planes to cover all hash
// functions (n_hash <= MISS_PLANES).
struct PendPool {
    unsigned long long m[64];
    unsigned long long miss[MISS_PLANES][64];   // the missing planes of the pool's words: read once per word, not once per occurrence
    int excl[64];
};

template <int S, int REC>
__global__ void __launch_bounds__(256) k_load_resolve_sm(const uint64_t* __restrict__ codes, uint64_t n_words, FdParams fp, Filt<REC> f,
                                                         uint32_t tb, const uint64_t* __restrict__ pending,
                                                         uint64_t plane_stride, unsigned long long* sure, DevCounters* cnt) {
    // Plain LDS objects indexed directly, so that the accesses are ds_read / ds_write: those execute in order for a wave, which is
    // what makes a word written by one lane visible to the lane that reads it next.  (Declared volatile, or reached through a
    // generic pointer, they become FLAT accesses, whose order between the lanes of a wave is NOT guaranteed.)  The wavefront-scope
    // fences around the refill keep the compiler from moving or caching the accesses across it.
    __shared__ PendPool pools[4];
    const int wid = (int)(threadIdx.x >> 6);
    const int lane = fd_lane();
    const uint64_t lt_mask = (1ULL << lane) - 1;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint64_t wv = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_groups = (n_words + 63) / 64;
    uint64_t grp = wv * n_groups / n_waves;
    const uint64_t grp_end = (wv + 1) * n_groups / n_waves;
    unsigned long long n_pass = 0;
    int next = 0, total = 0;
    uint64_t pool_word0 = 0;
    // S occurrences per lane (S independent first[] loads in flight): missing == 0 means the slot is free
    uint64_t hA[S], hB[S], item_p[S];
    uint32_t missing[S];
#pragma unroll
    for (int q = 0; q < S; q++) missing[q] = 0;
    for (;;) {
        uint64_t idle[S], any_idle = 0;
#pragma unroll
        for (int q = 0; q < S; q++) { idle[q] = __ballot(missing[q] == 0); any_idle |= idle[q]; }
        while (any_idle && (next < total || grp < grp_end)) {
            if (next == total) {
                const uint64_t w = grp * 64 + lane;
                const uint64_t m = w < n_words ? pending[w] : 0ULL;
                const int c = __popcll(m);
                int incl = c;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int o = __shfl_up(incl, d, 64);
                    if (lane >= d) incl += o;
                }
                pools[wid].m[lane] = m;
#pragma unroll
                for (int i = 0; i < MISS_PLANES; i++) pools[wid].miss[i][lane] = m ? pending[(i + 1) * plane_stride + w] : 0ULL;
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
            for (int q = 0; q < S; q++) {
                const int r = __popcll(idle[q] & lt_mask);
                const int avail = total - next;
                if (missing[q] == 0 && r < avail) {
                    const int t = next + r;
                    int sl = 0;
#pragma unroll
                    for (int step = 32; step > 0; step >>= 1)
                        if (pools[wid].excl[sl + step] <= t) sl += step;
                    const uint64_t w = pool_word0 + sl;
                    const int bit = select_bit(pools[wid].m[sl], t - pools[wid].excl[sl]);
                    item_p[q] = w * 64 + bit;
                    uint32_t ms = 0;
#pragma unroll
                    for (int i = 0; i < MISS_PLANES; i++) ms |= (uint32_t)((pools[wid].miss[i][sl] >> bit) & 1ULL) << i;
                    missing[q] = ms;             // pending => at least one bit was missing
                    fd_hash_pair(fd_canon(fd_kmer_at(codes, item_p[q], fp.k), fp.k), fp.tai_mask, hA[q], hB[q]);
                }
                next += min(__popcll(idle[q]), avail > 0 ? avail : 0);
            }
            any_idle = 0;
#pragma unroll
            for (int q = 0; q < S; q++) { idle[q] = __ballot(missing[q] == 0); any_idle |= idle[q]; }
        }
        bool mine = false;
#pragma unroll
        for (int q = 0; q < S; q++) mine |= missing[q] != 0;
        if (!__ballot(mine)) break;
        uint32_t seen[S];
#pragma unroll
        for (int q = 0; q < S; q++) {       // all S loads are issued before any result is looked at
            seen[q] = 0;
            if (missing[q]) {
                const uint64_t h = (hA[q] + (uint64_t)__builtin_ctz(missing[q]) * hB[q]) & fp.tai_mask;
                seen[q] = *f.time(h);
            }
        }
#pragma unroll
        for (int q = 0; q < S; q++) {
            if (!missing[q]) continue;
            if (!(seen[q] < tb + (uint32_t)item_p[q])) {
                missing[q] = 0;                               // not set before this occurrence: it stays out of bloo2
            } else {
                missing[q] &= missing[q] - 1;
                if (!missing[q]) {                            // rare: every bit was set earlier in this very batch
                    n_pass++;
                    uint64_t hh = hA[q];
                    for (int i = 0; i < fp.n_hash; i++) {
                        const uint32_t b = 1u << (hh & 31);
                        if (!(f.word(hh)[1] & b)) atomicOr(f.word(hh) + 1, b);
                        hh = (hh + hB[q]) & fp.tai_mask;
                    }
                    atomicOr(&sure[item_p[q] >> 6], 1ULL << (item_p[q] & 63));
                }
            }
        }
    }
    block_add(&cnt->to_bloo2, n_pass);
}

// ---- --mercy (utils/Bloom.cpp:300-333) -----------------------------------------------------------------------------
// With mercy the load also adds to bloo2 every run of low-coverage k-mers ("not contained in bloo1 when met") that sits
// between two solid ones, unless the solid k-mer next to the run looks like a junction in bloo1 (isJunction, :249-265).  bloo1
// evolves exactly as without mercy, and which occurrences were "contained" is the sure plane the two kernels above have
// just written; what is left is a small sequential state machine per unambiguous segment plus a few TIME-AWARE membership
// tests: bloo1 as it stood when occurrence t was processed = bits of the carried-in state or first set at a time <= t.
template <int REC>
__device__ __forceinline__ bool bloo1_contains_at(const Filt<REC>& f, uint64_t canon, uint32_t t, const FdParams& fp) {
    uint64_t hA, hB;
    fd_hash_pair(canon, fp.tai_mask, hA, hB);
    uint64_t h = hA;
    for (int i = 0; i < fp.n_hash; i++) {
        if (!((f.load(h).x >> (h & 31)) & 1u) && !(*f.time(h) <= t)) return false;
        h = (h + hB) & fp.tai_mask;
    }
    return true;
}

// isJunction(readKmer, bloo1, dir) as load_two_filters calls it: the cursor faces BACKWARD there, so the "real extension" is
// the reverse complement of the window before, whatever dir says; dir only picks the strand the four candidates extend.
template <int REC>
__device__ __forceinline__ bool mercy_is_junction(const uint64_t* __restrict__ codes, const Filt<REC>& f, uint32_t tb, uint64_t pos, bool dir_forward,
                                                  const FdParams& fp) {
    const uint64_t km = fd_kmer_at(codes, pos, fp.k), rc = fd_revcomp(km, fp.k);
    const uint64_t real_ext = ((rc << 2) | (uint64_t)(fd_base_at(codes, pos - 1) ^ 2)) & fp.kmask;
    const uint64_t from = dir_forward ? km : rc;
    for (int nt = 0; nt < 4; nt++) {
        const uint64_t e = ((from << 2) | (uint64_t)nt) & fp.kmask;
        if (e != real_ext && bloo1_contains_at(f, fd_canon(e, fp.k), tb + (uint32_t)pos, fp)) return true;
    }
    return false;
}

// one thread per 64-position word: the unambiguous segments (length >= k) that START in it
template <int REC>
__global__ void __launch_bounds__(256) k_load_mercy(const uint64_t* __restrict__ codes, const uint64_t* __restrict__ bad, uint64_t n_words,
                                                    FdParams fp, Filt<REC> f, uint32_t tb, const uint64_t* __restrict__ sure) {
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t good = ~bad[w];
        const uint64_t prev_good = w ? (~bad[w - 1]) >> 63 : 0;
        uint64_t starts = good & ~((good << 1) | prev_good);
        while (starts) {
            const uint64_t p = w * 64 + __builtin_ctzll(starts);
            starts &= starts - 1;
            uint64_t len = 0;                       // segment length: bad padding past the end terminates the scan
            for (;;) {
                const uint64_t v = fd_bits_at(bad, p + len);
                if (v) { len += __builtin_ctzll(v); break; }
                len += 64;
            }
            if (len < (uint64_t)fp.k) continue;
            const uint64_t n = len - fp.k + 1;      // windows p .. p+n-1, processed in this order (utils/Bloom.cpp:303)
            bool have_last = false;
            int64_t hv_lo = -1;                     // first window of the current hash_vals run, -1 = empty
            uint64_t sbits = 0;
            for (uint64_t i = 0; i < n; i++) {
                if ((i & 63) == 0) sbits = fd_bits_at(sure, p + i);
                const bool contained = (sbits >> (i & 63)) & 1ULL;
                const uint64_t pos = p + i;
                if (contained) {
                    have_last = true;
                    if (hv_lo >= 0) {               // came from low to high (:311-318)
                        if (!mercy_is_junction(codes, f, tb, pos, false, fp)) {
                            for (uint64_t q = p + (uint64_t)hv_lo; q < pos; q++) {
                                uint64_t hA, hB;
                                fd_hash_pair(fd_canon(fd_kmer_at(codes, q, fp.k), fp.k), fp.tai_mask, hA, hB);
                                uint64_t h = hA;
                                for (int b = 0; b < fp.n_hash; b++) {
                                    const uint32_t bit = 1u << (h & 31);
                                    if (!(f.word(h)[1] & bit)) atomicOr(f.word(h) + 1, bit);
                                    h = (h + hB) & fp.tai_mask;
                                }
                            }
                        }
                        hv_lo = -1;
                    }
                } else if (have_last && hv_lo < 0) {   // came from high to low (:322-326); later low k-mers just join the run
                    if (!mercy_is_junction(codes, f, tb, pos, true, fp)) hv_lo = (int64_t)i;
                }
            }
        }
    }
}

// pair[w] = {a[w], b[w]} (b == nullptr: zero) / the reverse / refresh of the carry half after a batch
__global__ void __launch_bounds__(256) k_pair_join(uint2* __restrict__ pair, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                                   uint64_t n32) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n32; i += (uint64_t)gridDim.x * blockDim.x)
        pair[i] = make_uint2(a[i], b ? b[i] : 0u);
}
__global__ void __launch_bounds__(256) k_pair_split(const uint2* __restrict__ pair, uint32_t* __restrict__ a, uint32_t* __restrict__ b,
                                                    uint64_t n32) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n32; i += (uint64_t)gridDim.x * blockDim.x) {
        uint2 v = pair[i];
        a[i] = v.x;
        b[i] = v.y;
    }
}
// the record layout's counterparts: a record starts as {carried-in bloo1 word, empty bloo2 word, ..., 32 times "never"}; lanes = the record's
// 64 words (the unused ones are written too: whole lines, no read-modify-write); at the end the two filter words go back to the .bloom arrays
__global__ void __launch_bounds__(256) k_rec_init(uint32_t* __restrict__ rec, const uint32_t* __restrict__ a, uint64_t n32) {
    const uint64_t total = n32 * REC_WORDS;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t w = (uint32_t)(i & (REC_WORDS - 1));
        if (w >= 48) continue;                             // the fourth line of a record is never read
        rec[i] = w == 0 ? a[i >> 6] : w == 1 ? 0u : 0xFFFFFFFFu;
    }
}
__global__ void __launch_bounds__(256) k_rec_split(const uint32_t* __restrict__ rec, uint32_t* __restrict__ a, uint32_t* __restrict__ b, uint64_t n32) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n32; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint2 v = *(const uint2*)(rec + i * REC_WORDS);
        a[i] = v.x;
        b[i] = v.y;
    }
}
// the sweep: 32 lanes per record read its 32 times (two coalesced lines), a ballot gives the record's "set since the last sweep" word
__global__ void __launch_bounds__(256) k_carry_from_rec(uint32_t* __restrict__ rec, uint64_t n32) {
    const uint64_t total = n32 * 32;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i >> 5;
        const uint32_t t = rec[r * REC_WORDS + 16 + (i & 31)];
        const uint64_t m = __ballot(t != 0xFFFFFFFFu);
        const uint32_t mine = (uint32_t)(fd_lane() < 32 ? m : m >> 32);
        if ((fd_lane() & 31) == 0 && mine) rec[r * REC_WORDS] |= mine;
    }
}

// carry := carry | bits set during the batch.  A bit is set iff its first-set time is no longer "never": one streaming
// pass over first[] (4 bytes per Bloom bit, lanes = consecutive bits, one ballot per 64 bits) replaces one atomicOr per
// newly set bit in the mark kernel.
__global__ void __launch_bounds__(256) k_carry_from_first(uint2* __restrict__ pair, const uint4* __restrict__ first4, uint64_t tai) {
    // 16 bytes per lane (a wave covers 256 consecutive Bloom bits = 1 KiB of first[]); the lane's four "was set" bits are
    // OR-reduced over groups of 16 lanes into one 64-bit word of the carry, which the group's first lane merges in
    const uint64_t n4 = tai / 4;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 v = first4[i];
        const uint32_t nib = (v.x != 0xFFFFFFFFu ? 1u : 0u) | (v.y != 0xFFFFFFFFu ? 2u : 0u) | (v.z != 0xFFFFFFFFu ? 4u : 0u) |
                             (v.w != 0xFFFFFFFFu ? 8u : 0u);
        const int g = fd_lane() & 15;
        uint32_t lo = g < 8 ? nib << (4 * g) : 0u, hi = g >= 8 ? nib << (4 * (g - 8)) : 0u;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            lo |= __shfl_xor(lo, o, 64);
            hi |= __shfl_xor(hi, o, 64);
        }
        if (g == 0) {
            const uint64_t w = i >> 3;   // 4 bits per lane: lane i's bits start at Bloom bit 4i = carry word 4i/32
            if (lo) pair[w].x |= lo;
            if (hi) pair[w + 1].x |= hi;
        }
    }
}

// carry |= bits of every occurrence that was NOT contained when met (those are exactly the occurrences that set bits in
// bloo1).  The alternative to the sweep when the filter is large: its cost follows the number of new k-mers of the batch
// (3 test-then-set accesses each), not the size of first[] (32 GiB per sweep for config 4's 2^33-bit filters).
template <int REC>
__global__ void __launch_bounds__(256) k_carry_set(const uint64_t* __restrict__ codes, const uint64_t* __restrict__ bad, uint64_t T,
                                                   uint64_t n_words, FdParams fp, Filt<REC> f, const uint64_t* __restrict__ sure) {
    const uint64_t total = n_words * 64;
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (uint64_t)gridDim.x * blockDim.x) {
        if (!(p < T && fd_window_ok(bad, p, fp.k)) || ((sure[p >> 6] >> (p & 63)) & 1ULL)) continue;
        uint64_t hA, hB;
        fd_hash_pair(fd_canon(fd_kmer_at(codes, p, fp.k), fp.k), fp.tai_mask, hA, hB);
        uint64_t h = hA;
        for (int i = 0; i < fp.n_hash; i++) {
            const uint32_t bit = 1u << (h & 31);
            if (!(f.word(h)[0] & bit)) atomicOr(f.word(h), bit);
            h = (h + hB) & fp.tai_mask;
        }
    }
}

// Read shards, fgpu_load_fixup: the occurrences the shard's own pass kept out of bloo2, looked at again with the lower shards' bits.
// An occurrence goes to bloo2 iff every one of its bits was set before it: by a lower shard (all of those come earlier in file order: the
// bit is in `prefix`) or earlier in this shard.  Which of its bits were NOT set earlier in this shard the own pass has written down when it
// resolved the occurrence (the `fail` planes of k_load_resolve, FGPU_LOAD_SHARD_PLANES) -- so the question left is whether those bits are all in
// the prefix.  No first-set time is read here: a shard may hold any number of positions (k_load_fixup_times below compares 32-bit times that
// count through the whole shard: at most 2^32 positions, i.e. config 4 from 8 GPUs on -- and costs the own pass nothing, so it stays for those).
__global__ void __launch_bounds__(256) k_load_fixup(const uint64_t* __restrict__ codes, const uint64_t* __restrict__ bad, uint64_t T,
                                                    uint64_t n_words, FdParams fp, const uint32_t* __restrict__ prefix,
                                                    const uint64_t* __restrict__ fail, uint64_t plane_stride, uint32_t* bloo2,
                                                    unsigned long long* sure, DevCounters* cnt) {
    unsigned long long n_pass = 0;
    const uint64_t total = n_words * 64;
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (uint64_t)gridDim.x * blockDim.x) {
        bool pass = false;
        if (p < T && fd_window_ok(bad, p, fp.k) && !((sure[p >> 6] >> (p & 63)) & 1ULL)) {
            uint32_t failed = 0;
#pragma unroll
            for (int i = 0; i < MISS_PLANES; i++) failed |= (uint32_t)((fail[i * plane_stride + (p >> 6)] >> (p & 63)) & 1ULL) << i;
            if (failed) {                               // (an occurrence outside bloo2 has at least one)
                uint64_t hA, hB;
                fd_hash_pair(fd_canon(fd_kmer_at(codes, p, fp.k), fp.k), fp.tai_mask, hA, hB);
                pass = true;
                uint64_t h = hA;
                for (int i = 0; i < fp.n_hash && pass; i++) {
                    if (((failed >> i) & 1u) && !((prefix[h >> 5] >> (h & 31)) & 1u)) pass = false;
                    h = (h + hB) & fp.tai_mask;
                }
                if (pass) {
                    n_pass++;
                    fd_bloom_set(bloo2, hA, hB, fp.tai_mask, fp.n_hash);
                }
            }
        }
        const uint64_t sm = __ballot(pass);   // lanes = the 64 positions of one plane word
        if (fd_lane() == 0 && sm) sure[p >> 6] |= sm;
    }
    block_add(&cnt->to_bloo2, n_pass);
}

// The same question answered from first-set times, for shards of fewer than 2^32 positions (FGPU_LOAD_SHARD_TIMES: the own pass dated every
// occurrence on one clock for the whole shard and resolved with the dense kernel, which costs it nothing; rounds 2-4's form): bit set before t
// <=> in the prefix or first set locally before t.
template <int REC>
__global__ void __launch_bounds__(256) k_load_fixup_times(const uint64_t* __restrict__ codes, const uint64_t* __restrict__ bad, uint64_t T,
                                                          uint64_t n_words, FdParams fp, const uint32_t* __restrict__ prefix,
                                                          Filt<REC> f, uint32_t tb, uint32_t* bloo2, unsigned long long* sure, DevCounters* cnt) {
    unsigned long long n_pass = 0;
    const uint64_t total = n_words * 64;
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (uint64_t)gridDim.x * blockDim.x) {
        bool pass = false;
        if (p < T && fd_window_ok(bad, p, fp.k) && !((sure[p >> 6] >> (p & 63)) & 1ULL)) {
            uint64_t hA, hB;
            fd_hash_pair(fd_canon(fd_kmer_at(codes, p, fp.k), fp.k), fp.tai_mask, hA, hB);
            pass = true;
            uint64_t h = hA;
            for (int i = 0; i < fp.n_hash; i++) {
                if (!((prefix[h >> 5] >> (h & 31)) & 1u) && !(*f.time(h) < tb + (uint32_t)p)) { pass = false; break; }
                h = (h + hB) & fp.tai_mask;
            }
            if (pass) {
                n_pass++;
                fd_bloom_set(bloo2, hA, hB, fp.tai_mask, fp.n_hash);
            }
        }
        const uint64_t sm = __ballot(pass);   // lanes = the 64 positions of one plane word
        if (fd_lane() == 0 && sm) sure[p >> 6] |= sm;
    }
    block_add(&cnt->to_bloo2, n_pass);
}

// multi-GPU helper: OR the bits of every k-mer into a bitmap, no ordering
__global__ void __launch_bounds__(256) k_presence(const uint64_t* __restrict__ codes, const uint64_t* __restrict__ bad,
                                                  uint64_t T, uint64_t n_words, FdParams fp, uint32_t* bitmap, DevCounters* cnt) {
    unsigned long long n_ok = 0;
    const uint64_t total = n_words * 64;
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (uint64_t)gridDim.x * blockDim.x) {
        if (!(p < T && fd_window_ok(bad, p, fp.k))) continue;
        n_ok++;
        uint64_t canon = fd_canon(fd_kmer_at(codes, p, fp.k), fp.k);
        uint64_t hA, hB;
        fd_hash_pair(canon, fp.tai_mask, hA, hB);
        fd_bloom_set(bitmap, hA, hB, fp.tai_mask, fp.n_hash);
    }
    block_add(&cnt->kmers, n_ok);
}

// unambiguous segments of length >= minlen (utils/Kmer.cpp:77; ReadScanner.cpp:268).  One thread per 64-position
// word of the bad mask: run starts are found with bit arithmetic, each start measures its run.
__global__ void __launch_bounds__(256) k_count_segments(const uint64_t* __restrict__ bad, uint64_t n_words, int minlen,
                                                        unsigned long long* out) {
    unsigned long long n = 0;
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t good = ~bad[w];
        uint64_t prev_good = w ? (~bad[w - 1]) >> 63 : 0;
        uint64_t starts = good & ~((good << 1) | prev_good);
        while (starts) {
            int s = __builtin_ctzll(starts);
            starts &= starts - 1;
            uint64_t p = w * 64 + s;
            int len = 0;
            while (len < minlen) {   // bad padding past the end terminates the scan
                uint64_t v = fd_bits_at(bad, p + len);
                if (v) { len += __builtin_ctzll(v); break; }
                len += 64;
            }
            if (len >= minlen) n++;
        }
    }
    block_add(out, n);
}

__global__ void __launch_bounds__(256) k_popcount(const uint4* __restrict__ words, uint64_t n16, unsigned long long* out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long acc = 0;
    for (; i < n16; i += stride) {
        uint4 v = words[i];
        acc += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
    }
    wave_add(out, acc);
}

__global__ void __launch_bounds__(256) k_bitmap_or(uint4* __restrict__ dst, const uint4* __restrict__ src, uint64_t n16) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n16; i += stride) {
        uint4 a = dst[i], b = src[i];
        a.x |= b.x; a.y |= b.y; a.z |= b.z; a.w |= b.w;
        dst[i] = a;
    }
}

__global__ void k_probe_hash(const uint64_t* __restrict__ kmers, uint64_t n, FdParams fp, uint64_t* canon, uint64_t* hA, uint64_t* hB) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t c = fd_canon(kmers[i], fp.k);
    uint64_t a, b;
    fd_hash_pair(c, fp.tai_mask, a, b);
    canon[i] = c;
    hA[i] = a;
    hB[i] = b;
}

__global__ void k_probe_contains(const uint32_t* __restrict__ bloom, const uint64_t* __restrict__ canon, uint64_t n, FdParams fp,
                                 unsigned char* out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = fd_bloom_contains_canon(bloom, canon[i], fp.tai_mask, fp.n_hash) ? 1 : 0;
}

}  // namespace

// fgpu_create touches one kernel of every translation unit from a helper thread: the runtime loads a unit's code object at the first use of
// one of its kernels (20-25 ms for the large units), which otherwise lands on the first batch of each pass
void fgpu_touch_load() {
    hipFuncAttributes attr;
    (void)hipFuncGetAttributes(&attr, (const void*)k_popcount);
}

// Forget the kept batches (their buffers are recycled by the next load pass).  keep_going: a new load pass starts.
void fgpu_resident_reset(fgpu_ctx* ctx, bool keep_going) {
    ctx->resident_count = 0;
    ctx->resident_bytes = 0;
    ctx->resident_open = keep_going && ctx->resident_budget > 0;
}

// Keep the batch's stream and its routed-to-bloo2 plane for the scan pass (data stays in HBM between the passes
// instead of being recomputed by probing); stops silently once the budget is used: the scan then probes as usual.
static int fgpu_resident_keep(fgpu_ctx* ctx) {
    if (!ctx->resident_open) return FGPU_OK;
    BatchBufs& bb = *ctx->cur;
    const uint64_t cb = 2 * (bb.n_words + FGPU_PADW) * 8, pb = (bb.n_words + FGPU_PADW) * 8;
    if (ctx->resident_bytes + cb + 2 * pb + (ctx->shard_planes ? MISS_PLANES * pb : 0) > ctx->resident_budget) {
        ctx->resident_open = false;   // batches pair by index: once one is missing, later ones would not line up
        return FGPU_OK;
    }
    if (ctx->resident_count == ctx->resident.size()) ctx->resident.push_back(new ResidentBatch());
    ResidentBatch& r = *ctx->resident[ctx->resident_count];
    const uint64_t fb = ctx->shard_planes ? MISS_PLANES * pb : 0;     // read shards: which bits of an occurrence were not set before it (fgpu_load_fixup)
    if (fgpu_ensure_b(ctx, &r.codes, cb) || fgpu_ensure_b(ctx, &r.bad, pb) || fgpu_ensure_b(ctx, &r.sure, pb) || (fb && fgpu_ensure_b(ctx, &r.fail, fb))) {
        (void)hipGetLastError();
        ctx->resident_open = false;   // out of memory: do without
        return FGPU_OK;
    }
    r.T = bb.T;
    r.n_words = bb.n_words;
    r.tb = ctx->cur_tb;
    FGPU_HIP(hipMemcpyAsync(r.codes.p, bb.codes.p, cb, hipMemcpyDeviceToDevice, ctx->stream));
    FGPU_HIP(hipMemcpyAsync(r.bad.p, bb.bad.p, pb, hipMemcpyDeviceToDevice, ctx->stream));
    FGPU_HIP(hipMemcpyAsync(r.sure.p, bb.sure.p, pb, hipMemcpyDeviceToDevice, ctx->stream));
    if (fb) FGPU_HIP(hipMemcpyAsync(r.fail.p, bb.fail.p, fb, hipMemcpyDeviceToDevice, ctx->stream));
    ctx->resident_count++;
    ctx->resident_bytes += cb + 2 * pb + fb;
    return FGPU_OK;
}

// carry |= bits set since the last sweep; closes the epoch (times start at 0 again: every bit with a time is now in the carry)
int fgpu_load_sweep(fgpu_ctx* ctx) {
    if (ctx->epoch_positions == 0) return FGPU_OK;
    if (ctx->rec_layout) FGPU_LAUNCH("carry_update", k_carry_from_rec, 8192, 256, ctx->rec, ctx->prm.tai / 32);
    else FGPU_LAUNCH("carry_update", k_carry_from_first, 4096, 256, ctx->pair, (const uint4*)ctx->first, ctx->prm.tai);
    ctx->swept_positions += ctx->epoch_positions;
    ctx->epoch_positions = 0;
    return FGPU_OK;
}

int fgpu_stage_load(fgpu_ctx* ctx) {
    BatchBufs& bb = *ctx->cur;
    if (bb.T == 0) return FGPU_OK;
    const uint64_t plane_stride = bb.n_words + FGPU_PADW;   // plane 0: pending; planes 1..MISS_PLANES: bit i missing from the carry
    int rc = fgpu_ensure_b(ctx, &bb.pending, (MISS_PLANES + 1) * plane_stride * 8);
    if (rc) return rc;
    if ((rc = fgpu_ensure_b(ctx, &bb.sure, (bb.n_words + FGPU_PADW) * 8))) return rc;
    const unsigned grid = fgpu_grid(bb.n_words * 64, 256);
    if ((rc = fgpu_util_count_segments(ctx, ctx->fd.k))) return rc;
    // Times are positions within the current EPOCH = the batches since the last sweep of first[] (k_carry_from_first).  A bit
    // that is still 0 in the carry has first[bit] == never or a time of this epoch, so the carry does not have to be brought up
    // to date after every batch: "set before t" = in the carry or first[bit] < t holds with any carry that is a subset of
    // bloo1 as of the epoch's start.  A lagging carry only sends more occurrences through the resolve kernel, and once the carry
    // holds a few coverages of the genome nearly every k-mer that will ever be in it already is: sweeps are made when an epoch
    // has grown to sweep_num/sweep_den of what the carry already covers (after batches 0, 1, 3, 7 ... of equal batches).
    const uint64_t span = bb.n_words * 64;
    if (!ctx->carry_by_set && !ctx->shard_times && ctx->epoch_positions + span >= 0xFFFFFFF0ULL && (rc = fgpu_load_sweep(ctx))) return rc;
    if (ctx->shard_times && ctx->pass_positions + span >= 0xFFFFFFF0ULL) {
        ctx->err = "FGPU_LOAD_SHARD_TIMES: the pass exceeds 2^32 stream positions (FGPU_LOAD_SHARD_PLANES has no such limit)";
        return FGPU_ERR_CAPACITY;
    }
    // FGPU_LOAD_SHARD_TIMES: one clock for the whole pass (sweeps still bring the carry up to date; they just do not restart it)
    const uint32_t tb = ctx->shard_times ? (uint32_t)ctx->pass_positions : ctx->carry_by_set ? 0u : (uint32_t)ctx->epoch_positions;
    // FGPU_LOAD_SHARD_PLANES: the resolve kernel also writes down WHICH bits of an occurrence were not set before it
    const bool keep_fail = ctx->shard_planes && ctx->fd.n_hash <= MISS_PLANES;
    if (keep_fail && (rc = fgpu_ensure_b(ctx, &bb.fail, MISS_PLANES * plane_stride * 8))) return rc;
    ctx->cur_tb = tb;
    ctx->pass_positions += span;
    static const int resolve_sm = getenv("FGPU_RESOLVE_SM") ? atoi(getenv("FGPU_RESOLVE_SM")) : 4096;
    static const int slots = getenv("FGPU_RESOLVE_SM_SLOTS") ? atoi(getenv("FGPU_RESOLVE_SM_SLOTS")) : 1;
    const unsigned rgrid = (unsigned)std::min<uint64_t>((bb.n_words + 255) / 256, (uint64_t)std::max(resolve_sm, 64));
    const bool mercy = (ctx->prm.flags & FGPU_FLAG_MERCY) != 0;
#define FGPU_LOAD_BATCH(REC, F)                                                                                                                       \
    do {                                                                                                                                              \
        FGPU_LAUNCH("load_mark", k_load_mark<REC>, grid, 256, (const uint64_t*)bb.codes.p, (const uint64_t*)bb.bad.p, bb.T, bb.n_words, ctx->fd, F,  \
                    tb, (uint64_t*)bb.pending.p, plane_stride, (uint64_t*)bb.sure.p, ctx->counters);                                                  \
        if (ctx->fd.n_hash <= MISS_PLANES && resolve_sm && !keep_fail) {                                                                              \
            if (slots <= 1)                                                                                                                            \
                FGPU_LAUNCH("load_resolve", (k_load_resolve_sm<1, REC>), rgrid, 256, (const uint64_t*)bb.codes.p, bb.n_words, ctx->fd, F, tb,          \
                            (const uint64_t*)bb.pending.p, plane_stride, (unsigned long long*)bb.sure.p, ctx->counters);                               \
            else if (slots == 2)                                                                                                                       \
                FGPU_LAUNCH("load_resolve", (k_load_resolve_sm<2, REC>), rgrid, 256, (const uint64_t*)bb.codes.p, bb.n_words, ctx->fd, F, tb,          \
                            (const uint64_t*)bb.pending.p, plane_stride, (unsigned long long*)bb.sure.p, ctx->counters);                               \
            else                                                                                                                                       \
                FGPU_LAUNCH("load_resolve", (k_load_resolve_sm<4, REC>), rgrid, 256, (const uint64_t*)bb.codes.p, bb.n_words, ctx->fd, F, tb,          \
                            (const uint64_t*)bb.pending.p, plane_stride, (unsigned long long*)bb.sure.p, ctx->counters);                               \
        } else {                                                                                                                                       \
            FGPU_LAUNCH("load_resolve", k_load_resolve<REC>, grid, 256, (const uint64_t*)bb.codes.p, bb.T, bb.n_words, ctx->fd, F, tb,                 \
                        (const uint64_t*)bb.pending.p, plane_stride, (uint64_t*)bb.sure.p, ctx->counters,                                              \
                        keep_fail ? (uint64_t*)bb.fail.p : (uint64_t*)nullptr);                                                                        \
        }                                                                                                                                              \
        if (mercy)                                                                                                                                     \
            FGPU_LAUNCH("load_mercy", k_load_mercy<REC>, fgpu_grid(bb.n_words, 256), 256, (const uint64_t*)bb.codes.p, (const uint64_t*)bb.bad.p,      \
                        bb.n_words, ctx->fd, F, tb, (const uint64_t*)bb.sure.p);                                                                       \
        if (ctx->carry_by_set)                                                                                                                         \
            FGPU_LAUNCH("carry_update", k_carry_set<REC>, grid, 256, (const uint64_t*)bb.codes.p, (const uint64_t*)bb.bad.p, bb.T, bb.n_words,         \
                        ctx->fd, F, (const uint64_t*)bb.sure.p);                                                                                       \
    } while (0)
    if (ctx->rec_layout) {
        const Filt<1> f = {ctx->rec, nullptr};
        FGPU_LOAD_BATCH(1, f);
    } else {
        const Filt<0> f = {(uint32_t*)ctx->pair, ctx->first};
        FGPU_LOAD_BATCH(0, f);
    }
#undef FGPU_LOAD_BATCH
    // carry := carry | bits set during this batch -- or later: the carry may lag behind (see fgpu_load_sweep)
    if (!ctx->carry_by_set) {
        ctx->epoch_positions += span;
        if (ctx->epoch_positions * ctx->sweep_den >= ctx->swept_positions * ctx->sweep_num && ctx->epoch_positions >= ctx->sweep_min &&
            (rc = fgpu_load_sweep(ctx))) return rc;
    }
    return fgpu_resident_keep(ctx);
}

// interleave the carried-in bloo1 with an empty bloo2 at the start of a load pass, split them again at its end
int fgpu_load_pair_begin(fgpu_ctx* ctx) {
    if (ctx->rec_layout) FGPU_LAUNCH("pair_join", k_rec_init, 8192, 256, ctx->rec, (const uint32_t*)ctx->bloo1, ctx->bloom_bytes / 4);
    else FGPU_LAUNCH("pair_join", k_pair_join, 2048, 256, ctx->pair, (const uint32_t*)ctx->bloo1, (const uint32_t*)nullptr, ctx->bloom_bytes / 4);
    return FGPU_OK;
}
int fgpu_load_pair_end(fgpu_ctx* ctx) {
    if (ctx->rec_layout) FGPU_LAUNCH("pair_split", k_rec_split, 4096, 256, (const uint32_t*)ctx->rec, ctx->bloo1, ctx->bloo2, ctx->bloom_bytes / 4);
    else FGPU_LAUNCH("pair_split", k_pair_split, 2048, 256, (const uint2*)ctx->pair, ctx->bloo1, ctx->bloo2, ctx->bloom_bytes / 4);
    return FGPU_OK;
}

int fgpu_stage_fixup(fgpu_ctx* ctx, const uint32_t* prefix) {
    for (uint64_t i = 0; i < ctx->resident_count; i++) {
        ResidentBatch& r = *ctx->resident[i];
        if (!r.T) continue;
        if (ctx->shard_planes) {
            FGPU_LAUNCH("load_fixup", k_load_fixup, fgpu_grid(r.n_words * 64, 256), 256, (const uint64_t*)r.codes.p, (const uint64_t*)r.bad.p, r.T, r.n_words,
                        ctx->fd, prefix, (const uint64_t*)r.fail.p, r.n_words + FGPU_PADW, ctx->bloo2, (unsigned long long*)r.sure.p, ctx->counters);
        } else if (ctx->rec_layout) {
            const Filt<1> f = {ctx->rec, nullptr};
            FGPU_LAUNCH("load_fixup", k_load_fixup_times<1>, fgpu_grid(r.n_words * 64, 256), 256, (const uint64_t*)r.codes.p, (const uint64_t*)r.bad.p, r.T,
                        r.n_words, ctx->fd, prefix, f, r.tb, ctx->bloo2, (unsigned long long*)r.sure.p, ctx->counters);
        } else {
            const Filt<0> f = {(uint32_t*)ctx->pair, ctx->first};
            FGPU_LAUNCH("load_fixup", k_load_fixup_times<0>, fgpu_grid(r.n_words * 64, 256), 256, (const uint64_t*)r.codes.p, (const uint64_t*)r.bad.p, r.T,
                        r.n_words, ctx->fd, prefix, f, r.tb, ctx->bloo2, (unsigned long long*)r.sure.p, ctx->counters);
        }
    }
    return FGPU_OK;
}

int fgpu_stage_presence(fgpu_ctx* ctx) {
    BatchBufs& bb = *ctx->cur;
    if (bb.T == 0) return FGPU_OK;
    int rc;
    if ((rc = fgpu_util_count_segments(ctx, ctx->fd.k))) return rc;
    FGPU_LAUNCH("presence", k_presence, fgpu_grid(bb.n_words * 64, 256), 256, (const uint64_t*)bb.codes.p, (const uint64_t*)bb.bad.p,
                bb.T, bb.n_words, ctx->fd, ctx->bloo1, ctx->counters);
    return FGPU_OK;
}

// ---- small utilities used by api.hip and scan_pure.hip --------------------------------------------
int fgpu_util_count_segments(fgpu_ctx* ctx, int minlen) {
    BatchBufs& bb = *ctx->cur;
    FGPU_LAUNCH("count_segments", k_count_segments, std::min(fgpu_grid(bb.n_words, 256), 256u), 256, (const uint64_t*)bb.bad.p, bb.n_words, minlen,
                &ctx->counters->segments);
    return FGPU_OK;
}

int fgpu_util_popcount(fgpu_ctx* ctx, const void* dev, uint64_t nbytes, unsigned long long* dev_out) {
    FGPU_LAUNCH("popcount", k_popcount, 1024, 256, (const uint4*)dev, nbytes / 16, dev_out);
    return FGPU_OK;
}

int fgpu_util_or(fgpu_ctx* ctx, void* dst, const void* src, uint64_t nbytes) {
    FGPU_LAUNCH("bitmap_or", k_bitmap_or, 2048, 256, (uint4*)dst, (const uint4*)src, nbytes / 16);
    return FGPU_OK;
}

int fgpu_util_probe_hash(fgpu_ctx* ctx, const uint64_t* d_kmers, uint64_t n, uint64_t* d_canon, uint64_t* d_hA, uint64_t* d_hB) {
    FGPU_LAUNCH("probe_hash", k_probe_hash, fgpu_blocks(n, 256), 256, d_kmers, n, ctx->fd, d_canon, d_hA, d_hB);
    return FGPU_OK;
}

int fgpu_util_probe_contains(fgpu_ctx* ctx, const uint32_t* bloom, const uint64_t* d_canon, uint64_t n, unsigned char* d_out) {
    FGPU_LAUNCH("probe_contains", k_probe_contains, fgpu_blocks(n, 256), 256, bloom, d_canon, n, ctx->fd, d_out);
    return FGPU_OK;
}
