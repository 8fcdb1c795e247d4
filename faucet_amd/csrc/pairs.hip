// pairs.hip — the long pair filter on the device: scanReads' paired-end loop (src/ReadScanner.cpp:317-343) over the lists
// scanInputRead returns (the stops the harvest builds, scan_walk.hip), with Bloom::containsPair / addPair (utils/Bloom.cpp:127-154).
//
// The reference, per read pair (records 2p and 2p+1 of the scan file), when both lists are non-empty:
//     for pair1 in list(2p):   if no pair2 in list(2p+1) has containsPair(pair1, pair2):   addPair(pair1, list(2p+1).front())
// Check-then-insert in file order: whether an element inserts depends on what earlier elements inserted.  The exact parallel form is the one
// pass 1 uses, with one difference -- an element that finds a partner inserts NOTHING, so the set of inserters is
// itself the unknown:
//   * every (read pair, pair1) element is an ITEM with a time t = its index in the batch's list array (file order);
//   * filter state seen by item t = carried-in filter (all earlier batches)  OR  bits of the inserts of items < t of this batch, i.e.
//       bit b set before t  <=>  b in carry  or  first[b] < t,      first[b] = min { t' : item t' inserts and b is a bit of its pair };
//   * round 0: an item that finds a partner against the carry alone is PAIRED for good (bits are only ever set); every other item is
//     assumed to insert and posts atomicMin(first[b], t) for the bits of (pair1, front) that the carry lacks;
//   * round r: every undecided item evaluates its checks against (carry, first[]); items whose answer differs from their assumption flip,
//     first[] is rebuilt from the new set of inserters, and the round repeats until nothing flips.
// The earliest item whose assumption is wrong sees only correct earlier inserts (a later item's time never passes the `< t` test of an
// earlier one), so every round settles it and everything before it: the fixed point is unique and is the sequential run's, reached after at
// most (longest chain of flips) + 1 rounds -- two or three on real lists.  Then the inserters' bits are ORed into the filter, which is the
// carry of the next batch.  A first end whose mate opens the next batch waits in `kept`.
#include <algorithm>

#include "fgpu_ctx.h"

namespace {

constexpr uint32_t LP_NEVER = 0xFFFFFFFFu;
enum : uint8_t { LP_NONE = 0, LP_FINAL = 1, LP_INSERT = 2, LP_PAIRED = 3 };

struct LpFilter {
    uint32_t* bits;       // the filter: tai / 8 bytes, bit p = bit (p & 31) of word p >> 5 (utils/Bloom.h:44-53 on little-endian words)
    uint32_t* first;      // first-set time per filter bit of the batch in hand; LP_NEVER between batches
    uint64_t mask;        // tai - 1
    int n_hash;
};

struct LpLists {
    const uint64_t* canon;   // canonical k-mer of every list element (all a JuncPair is hashed by)
    const uint64_t* h0;      // oldHash(canon, seed 0) & mask
    const uint64_t* h1;      // oldHash(canon, seed 1) & mask
    const uint32_t* vread;   // virtual read of every element (a waiting first end is virtual read 0)
    const uint32_t* rs;      // first element of every virtual read, n_vreads + 1 entries
    uint32_t n_elems, n_vreads;
};

// hA, hB of JuncPair(a, b): the smaller canonical k-mer under seed 0, the larger under seed 1 (utils/Bloom.cpp:127-154)
__device__ __forceinline__ void lp_pair_hash(const LpLists& L, uint32_t a, uint32_t b, uint64_t& hA, uint64_t& hB) {
    const bool a_small = L.canon[a] <= L.canon[b];
    hA = a_small ? L.h0[a] : L.h0[b];
    hB = a_small ? L.h1[b] : L.h1[a];
}

// canonical form and hashes of a batch's stops, behind the `n_kept` elements of a waiting first end
__global__ void __launch_bounds__(256) k_lp_prepare(const fgpu_stop* __restrict__ stops, uint32_t n_stops, uint32_t n_kept, uint32_t vshift, int k,
                                                    uint64_t mask, uint64_t* __restrict__ canon, uint64_t* __restrict__ h0,
                                                    uint64_t* __restrict__ h1, uint32_t* __restrict__ vread) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_kept) vread[i] = 0;
    if (i >= n_stops) return;
    const uint64_t c = fd_canon(stops[i].ext, k);
    canon[n_kept + i] = c;
    h0[n_kept + i] = fd_old_hash(c, FD_SEED0) & mask;
    h1[n_kept + i] = fd_old_hash(c, FD_SEED1) & mask;
    vread[n_kept + i] = stops[i].read + vshift;
}

// rs[r] = first element whose virtual read is >= r (elements are sorted by read); rs[n_vreads] = n_elems
__global__ void __launch_bounds__(256) k_lp_read_starts(const uint32_t* __restrict__ vread, uint32_t n_elems, uint32_t n_vreads, uint32_t* __restrict__ rs) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_vreads) return;
    uint32_t lo = 0, hi = n_elems;
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (vread[mid] < r) lo = mid + 1; else hi = mid;
    }
    rs[r] = lo;
}

// "Empty count / not empty count" (src/ReadScanner.cpp:318-342): per complete read pair of the batch
__global__ void __launch_bounds__(256) k_lp_count(const uint32_t* __restrict__ rs, uint32_t n_pairs, unsigned long long* __restrict__ counts) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    bool both = false;
    if (p < n_pairs) both = rs[2 * p + 1] > rs[2 * p] && rs[2 * p + 2] > rs[2 * p + 1];
    const unsigned long long m = __ballot(both), in_range = __ballot(p < n_pairs);
    if ((threadIdx.x & 63) == 0) {
        if (m) atomicAdd(&counts[1], (unsigned long long)__popcll(m));
        if (in_range & ~m) atomicAdd(&counts[0], (unsigned long long)__popcll(in_range & ~m));
    }
}

__device__ __forceinline__ bool lp_carry_bit(const LpFilter& F, uint64_t h) { return (F.bits[h >> 5] >> (h & 31)) & 1u; }

// the element's read pair: [b0, b1) = the second end's list.  false: the element is no item (second end, empty mate, waiting first end)
__device__ __forceinline__ bool lp_item(const LpLists& L, uint32_t e, uint32_t& b0, uint32_t& b1) {
    const uint32_t r = L.vread[e];
    if ((r & 1u) || r + 1 >= L.n_vreads) return false;
    b0 = L.rs[r + 1];
    b1 = L.rs[r + 2];
    return b1 > b0;
}

// round 0: paired against the carried-in filter alone?  Otherwise assume "insert" and post the times of the missing bits.
__global__ void __launch_bounds__(256) k_lp_init(LpLists L, LpFilter F, uint8_t* __restrict__ state, uint8_t* __restrict__ state_new,
                                                 unsigned long long* __restrict__ diag) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t b0, b1;
    uint8_t st = LP_NONE;
    if (e < L.n_elems && lp_item(L, e, b0, b1)) {
        bool paired = false;
        for (uint32_t j = b0; j < b1 && !paired; j++) {
            uint64_t hA, hB;
            lp_pair_hash(L, e, j, hA, hB);
            bool all = true;
            for (int i = 0; i < F.n_hash && all; i++) { all = lp_carry_bit(F, hA); hA = (hA + hB) & F.mask; }
            paired = all;
        }
        st = paired ? LP_FINAL : LP_INSERT;
        if (!paired) {
            uint64_t hA, hB;
            lp_pair_hash(L, e, b0, hA, hB);
            for (int i = 0; i < F.n_hash; i++) {
                if (!lp_carry_bit(F, hA)) atomicMin(&F.first[hA], e);
                hA = (hA + hB) & F.mask;
            }
        }
    }
    if (e < L.n_elems) {
        state[e] = st;
        state_new[e] = st;
    }
    // [0] undecided after round 0, [1] paired by the carry: one atomic per wave and counter (a same-address atomic per item serialises: 9.3 M of
    // them were 96 ms of config 3's scan)
    const unsigned long long undecided = __ballot(st == LP_INSERT), final_ = __ballot(st == LP_FINAL);
    if ((threadIdx.x & 63) == 0) {
        if (undecided) atomicAdd(&diag[0], (unsigned long long)__popcll(undecided));
        if (final_) atomicAdd(&diag[1], (unsigned long long)__popcll(final_));
    }
}

// round r >= 1: does the item find a partner in (carry, inserts of earlier items as currently assumed)?
__global__ void __launch_bounds__(256) k_lp_eval(LpLists L, LpFilter F, const uint8_t* __restrict__ state, uint8_t* __restrict__ state_new,
                                                 const uint32_t* __restrict__ flips_before, uint32_t* __restrict__ flips) {
    if (flips_before && *flips_before == 0) return;   // settled in an earlier round of this group of launches
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= L.n_elems) return;
    const uint8_t st = state[e];
    if (st != LP_INSERT && st != LP_PAIRED) return;
    const uint32_t r = L.vread[e], b0 = L.rs[r + 1], b1 = L.rs[r + 2];
    bool paired = false;
    for (uint32_t j = b0; j < b1 && !paired; j++) {
        uint64_t hA, hB;
        lp_pair_hash(L, e, j, hA, hB);
        bool all = true;
        for (int i = 0; i < F.n_hash && all; i++) {
            all = lp_carry_bit(F, hA) || F.first[hA] < e;
            hA = (hA + hB) & F.mask;
        }
        paired = all;
    }
    const uint8_t now = paired ? LP_PAIRED : LP_INSERT;
    state_new[e] = now;
    if (now != st) {                       // only "any" is asked: one atomic per wave that holds a flip
        const unsigned long long m = __ballot(true);
        if ((threadIdx.x & 63) == (unsigned)__builtin_ctzll(m)) atomicOr(flips, 1u);
    }
}

// first[] is rebuilt for the new assumptions: the bits of every item that inserted under the old ones are taken back ...
__global__ void __launch_bounds__(256) k_lp_withdraw(LpLists L, LpFilter F, const uint8_t* __restrict__ state, const uint32_t* __restrict__ flips) {
    if (*flips == 0) return;
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= L.n_elems || state[e] != LP_INSERT) return;
    uint64_t hA, hB;
    lp_pair_hash(L, e, L.rs[L.vread[e] + 1], hA, hB);
    for (int i = 0; i < F.n_hash; i++) { F.first[hA] = LP_NEVER; hA = (hA + hB) & F.mask; }
}
// ... and those of every item that inserts under the new ones are posted; the new assumptions become the current ones
__global__ void __launch_bounds__(256) k_lp_post(LpLists L, LpFilter F, uint8_t* __restrict__ state, const uint8_t* __restrict__ state_new,
                                                 const uint32_t* __restrict__ flips) {
    if (*flips == 0) return;
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= L.n_elems) return;
    const uint8_t now = state_new[e];
    state[e] = now;
    if (now != LP_INSERT) return;
    uint64_t hA, hB;
    lp_pair_hash(L, e, L.rs[L.vread[e] + 1], hA, hB);
    for (int i = 0; i < F.n_hash; i++) {
        if (!lp_carry_bit(F, hA)) atomicMin(&F.first[hA], e);
        hA = (hA + hB) & F.mask;
    }
}

// settled: the inserters' bits go into the filter (the carry of the next batch) and their times out of first[]
__global__ void __launch_bounds__(256) k_lp_commit(LpLists L, LpFilter F, const uint8_t* __restrict__ state, unsigned long long* __restrict__ diag,
                                                   const uint32_t* __restrict__ last_flips) {
    if (last_flips && *last_flips != 0) return;      // the rounds issued ahead did not settle: the host issues more, then commits (lp_close)
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    const bool ins = e < L.n_elems && state[e] == LP_INSERT;
    if (ins) {
        uint64_t hA, hB;
        lp_pair_hash(L, e, L.rs[L.vread[e] + 1], hA, hB);
        for (int i = 0; i < F.n_hash; i++) {
            atomicOr(&F.bits[hA >> 5], 1u << (hA & 31));
            F.first[hA] = LP_NEVER;
            hA = (hA + hB) & F.mask;
        }
    }
    const unsigned long long m = __ballot(ins);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&diag[2], (unsigned long long)__popcll(m));   // addPair calls
}

}  // namespace

// The rounds of one batch are issued AHEAD -- kAhead of them, each a no-op once an earlier one has settled -- with the commit behind them, and the
// host looks at the outcome when the next batch of lists arrives (or the scan ends): no wait of the host per batch (round 5; there was one per
// group of four rounds and one per odd batch).  lp_close is that look: how many rounds the batch took, more rounds and the commit if kAhead did
// not settle it (not seen so far: 4 at most on config 3's shape), and the list of a first end that waits for its mate.
constexpr int kAhead = 8;

static int lp_rounds(fgpu_ctx* ctx, const LpLists& L, const LpFilter& F, uint8_t* state, uint8_t* state_new, uint32_t* d_flips, int n, unsigned blocks) {
    FGPU_HIP(hipMemsetAsync(d_flips, 0, 4 * n, ctx->stream));
    for (int g = 0; g < n; g++) {
        FGPU_LAUNCH("long_pairs", k_lp_eval, blocks, 256, L, F, (const uint8_t*)state, state_new, g ? (const uint32_t*)(d_flips + g - 1) : (const uint32_t*)nullptr, d_flips + g);
        FGPU_LAUNCH("long_pairs", k_lp_withdraw, blocks, 256, L, F, (const uint8_t*)state, (const uint32_t*)(d_flips + g));
        FGPU_LAUNCH("long_pairs", k_lp_post, blocks, 256, L, F, state, (const uint8_t*)state_new, (const uint32_t*)(d_flips + g));
    }
    return FGPU_OK;
}

// the outcome of the batch whose rounds are in flight; everything it queued must have run (the callers have synchronised the stream, or do here)
int fgpu_long_pairs_close(fgpu_ctx* ctx) {
    LongPairs& lp = ctx->lp;
    if (!lp.open) return FGPU_OK;
    lp.open = false;
    if (hipEventQuery(lp.ev_open) != hipSuccess) FGPU_HIP(fgpu_sync_event(ctx, lp.ev_open));
    const uint32_t n_elems = lp.open_elems, n_vreads = lp.open_vreads;
    uint64_t *canon = (uint64_t*)lp.canon.p, *h0 = (uint64_t*)lp.h0.p, *h1 = (uint64_t*)lp.h1.p;
    if (lp.open_filter) {
        uint8_t *state = (uint8_t*)lp.state.p, *state_new = state + n_elems;
        const LpLists L = {canon, h0, h1, (uint32_t*)lp.vread.p, (uint32_t*)lp.rs.p, n_elems, n_vreads};
        const LpFilter F = {lp.bits, lp.first, lp.tai ? lp.tai - 1 : 0, lp.n_hash};
        unsigned long long* d_diag = (unsigned long long*)lp.dev.p;
        uint32_t* d_flips = (uint32_t*)((unsigned long long*)lp.dev.p + 8);
        const unsigned blocks = fgpu_blocks(std::max<uint64_t>(n_elems, 1), 256);
        int used = kAhead;
        for (int g = 0; g < kAhead; g++)
            if (lp.flips_host[g] == 0) { used = g + 1; break; }
        uint64_t rounds = (uint64_t)used;
        if (lp.flips_host[kAhead - 1] != 0) {          // not settled by the rounds issued ahead: the old way, a look per group, then the commit
            constexpr int kGroup = 4;
            for (;;) {
                if (int rc = lp_rounds(ctx, L, F, state, state_new, d_flips, kGroup, blocks)) return rc;
                FGPU_HIP(hipMemcpyAsync(lp.flips_host, d_flips, 4 * kGroup, hipMemcpyDeviceToHost, ctx->stream));
                FGPU_HIP(fgpu_sync_stream(ctx, ctx->stream));
                int u = kGroup;
                for (int g = 0; g < kGroup; g++)
                    if (lp.flips_host[g] == 0) { u = g + 1; break; }
                rounds += (uint64_t)u;
                if (lp.flips_host[u - 1] == 0) break;
                if (rounds > (uint64_t)n_elems + kAhead + kGroup) { ctx->err = "long pair filter: the rounds did not settle (internal error)"; return FGPU_ERR_STATE; }
            }
            FGPU_LAUNCH("long_pairs", k_lp_commit, blocks, 256, L, F, (const uint8_t*)state, d_diag, (const uint32_t*)nullptr);
        }
        lp.rounds += rounds;
        lp.max_rounds = std::max<uint64_t>(lp.max_rounds, rounds);
    }
    // a first end at the end of the batch waits for its mate: its list is kept (canonical forms and hashes)
    uint32_t new_kept = 0;
    if (lp.open_odd) {
        const uint32_t last_start = lp.flips_host[kAhead];
        new_kept = n_elems - last_start;
        if (new_kept) {
            DevBuf& nk = lp.kept_set[lp.kept_cur ^ 1];
            if (int rc = fgpu_ensure_b(ctx, &nk, 24ULL * new_kept)) return rc;
            uint64_t* kp = (uint64_t*)nk.p;
            FGPU_HIP(hipMemcpyAsync(kp, canon + last_start, 8ULL * new_kept, hipMemcpyDeviceToDevice, ctx->stream));
            FGPU_HIP(hipMemcpyAsync(kp + new_kept, h0 + last_start, 8ULL * new_kept, hipMemcpyDeviceToDevice, ctx->stream));
            FGPU_HIP(hipMemcpyAsync(kp + 2 * new_kept, h1 + last_start, 8ULL * new_kept, hipMemcpyDeviceToDevice, ctx->stream));
            lp.kept_cur ^= 1;
        }
    }
    lp.pending_first = lp.open_odd;
    lp.n_kept = new_kept;
    return FGPU_OK;
}

// Applies scanReads' paired-end loop to the lists of one harvested batch (device array of n_stops stops, sorted by read; n_reads reads).
// Called by fgpu_scan_harvest, in scan order, on the main stream.
int fgpu_long_pairs_batch(fgpu_ctx* ctx, const fgpu_stop* d_stops, uint64_t n_stops, uint64_t n_reads) {
    LongPairs& lp = ctx->lp;
    if (!lp.mode || !n_reads) return FGPU_OK;
    if (int rc = fgpu_long_pairs_close(ctx)) return rc;     // the batch before: its waiting first end, its rounds
    const uint64_t n_elems64 = lp.n_kept + n_stops, n_vreads64 = n_reads + (lp.pending_first ? 1 : 0);
    if (n_elems64 >= 0xFFFFFFF0ULL || n_vreads64 >= 0xFFFFFFF0ULL) { ctx->err = "long pair filter: more than 2^32 list elements in one batch"; return FGPU_ERR_CAPACITY; }
    const uint32_t n_elems = (uint32_t)n_elems64, n_vreads = (uint32_t)n_vreads64, n_pairs = n_vreads / 2, n_kept = (uint32_t)lp.n_kept;
    const bool odd = n_vreads & 1u;
    lp.batches++;
    if (!n_elems) {   // nothing but empty lists: every complete pair counts as empty, a first end without elements may be left waiting
        lp.empty_host += n_pairs;
        lp.pending_first = odd;
        lp.n_kept = 0;
        return FGPU_OK;
    }
    int rc;
    if ((rc = fgpu_ensure_b(ctx, &lp.canon, 8ULL * n_elems)) || (rc = fgpu_ensure_b(ctx, &lp.h0, 8ULL * n_elems)) || (rc = fgpu_ensure_b(ctx, &lp.h1, 8ULL * n_elems)) ||
        (rc = fgpu_ensure_b(ctx, &lp.vread, 4ULL * n_elems)) || (rc = fgpu_ensure_b(ctx, &lp.rs, 4ULL * (n_vreads + 2))) ||
        (rc = fgpu_ensure_b(ctx, &lp.state, 2ULL * n_elems)))
        return rc;
    uint64_t *canon = (uint64_t*)lp.canon.p, *h0 = (uint64_t*)lp.h0.p, *h1 = (uint64_t*)lp.h1.p;
    uint32_t *vread = (uint32_t*)lp.vread.p, *rs = (uint32_t*)lp.rs.p;
    uint8_t *state = (uint8_t*)lp.state.p, *state_new = state + n_elems;
    if (n_kept) {   // the waiting first end's list comes first
        const uint64_t* kept = (const uint64_t*)lp.kept_set[lp.kept_cur].p;
        FGPU_HIP(hipMemcpyAsync(canon, kept, 8ULL * n_kept, hipMemcpyDeviceToDevice, ctx->stream));
        FGPU_HIP(hipMemcpyAsync(h0, kept + n_kept, 8ULL * n_kept, hipMemcpyDeviceToDevice, ctx->stream));
        FGPU_HIP(hipMemcpyAsync(h1, kept + 2 * n_kept, 8ULL * n_kept, hipMemcpyDeviceToDevice, ctx->stream));
    }
    const uint64_t mask = lp.tai ? lp.tai - 1 : 0;
    const unsigned blocks = fgpu_blocks(std::max<uint64_t>(n_elems, 1), 256);
    FGPU_LAUNCH("long_pairs", k_lp_prepare, fgpu_blocks(std::max<uint32_t>((uint32_t)n_stops, n_kept), 256), 256, d_stops, (uint32_t)n_stops, n_kept,
                lp.pending_first ? 1u : 0u, ctx->fd.k, mask, canon, h0, h1, vread);
    FGPU_LAUNCH("long_pairs", k_lp_read_starts, fgpu_blocks(n_vreads + 1, 256), 256, (const uint32_t*)vread, n_elems, n_vreads, rs);
    unsigned long long* d_diag = (unsigned long long*)lp.dev.p;          // [0..2] diagnostics, [3] empty, [4] not empty
    uint32_t* d_flips = (uint32_t*)((unsigned long long*)lp.dev.p + 8);  // one counter per round issued ahead
    if (n_pairs) FGPU_LAUNCH("long_pairs", k_lp_count, fgpu_blocks(n_pairs, 256), 256, (const uint32_t*)rs, n_pairs, d_diag + 3);
    const LpLists L = {canon, h0, h1, vread, rs, n_elems, n_vreads};
    const bool filter = lp.mode == FGPU_LONG_PAIRS_FILTER && n_pairs;
    if (filter) {
        const LpFilter F = {lp.bits, lp.first, mask, lp.n_hash};
        FGPU_LAUNCH("long_pairs", k_lp_init, blocks, 256, L, F, state, state_new, d_diag);
        if ((rc = lp_rounds(ctx, L, F, state, state_new, d_flips, kAhead, blocks))) return rc;
        FGPU_LAUNCH("long_pairs", k_lp_commit, blocks, 256, L, F, (const uint8_t*)state, d_diag, (const uint32_t*)(d_flips + kAhead - 1));
        FGPU_HIP(hipMemcpyAsync(lp.flips_host, d_flips, 4 * kAhead, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (odd) FGPU_HIP(hipMemcpyAsync(lp.flips_host + kAhead, rs + (n_vreads - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
    if (!filter && !odd) {           // nothing to look at later
        lp.pending_first = false;
        lp.n_kept = 0;
        return FGPU_OK;
    }
    FGPU_HIP(hipEventRecord(lp.ev_open, ctx->stream));
    lp.open = true;
    lp.open_elems = n_elems;
    lp.open_vreads = n_vreads;
    lp.open_odd = odd;
    lp.open_filter = filter;
    return FGPU_OK;
}

// a scan starts with an empty filter, no waiting first end and zeroed counts
int fgpu_long_pairs_reset(fgpu_ctx* ctx) {
    LongPairs& lp = ctx->lp;
    if (!lp.mode) return FGPU_OK;
    lp.open = false;                 // (a scan that was abandoned with a batch in flight: its outcome is of no interest any more)
    if (lp.bits) FGPU_HIP(hipMemsetAsync(lp.bits, 0, lp.tai / 8, ctx->stream));
    if (lp.first) FGPU_HIP(hipMemsetAsync(lp.first, 0xFF, lp.tai * 4, ctx->stream));
    FGPU_HIP(hipMemsetAsync(lp.dev.p, 0, 128, ctx->stream));
    lp.pending_first = false;
    lp.n_kept = 0;
    lp.empty_host = lp.rounds = lp.max_rounds = lp.batches = 0;
    return FGPU_OK;
}

extern "C" {

int fgpu_scan_long_pairs(fgpu_ctx* ctx, uint64_t tai, int32_t n_hash, int32_t mode) {
    if (!ctx) return FGPU_ERR_ARG;
    if (ctx->phase != 0) { ctx->err = "fgpu_scan_long_pairs while a pass is open"; return FGPU_ERR_STATE; }
    FGPU_HIP(hipSetDevice(ctx->prm.device));
    LongPairs& lp = ctx->lp;
    if (lp.bits || lp.first) {
        FGPU_HIP(fgpu_sync_stream(ctx, ctx->stream));
        if (lp.bits) FGPU_HIP(hipFree(lp.bits));
        if (lp.first) FGPU_HIP(hipFree(lp.first));
        lp.bits = lp.first = nullptr;
    }
    lp.mode = FGPU_LONG_PAIRS_OFF;
    lp.tai = 0;
    if (mode == FGPU_LONG_PAIRS_OFF) return FGPU_OK;
    if (mode != FGPU_LONG_PAIRS_COUNT && mode != FGPU_LONG_PAIRS_FILTER) { ctx->err = "fgpu_scan_long_pairs: mode must be FGPU_LONG_PAIRS_OFF / _COUNT / _FILTER"; return FGPU_ERR_ARG; }
    if (!ctx->record_stops) { ctx->err = "fgpu_scan_long_pairs needs FGPU_FLAG_RECORD_STOPS"; return FGPU_ERR_STATE; }
    if (mode == FGPU_LONG_PAIRS_FILTER) {
        if (!tai || (tai & (tai - 1)) || tai < 128 || n_hash < 1 || n_hash > 32) { ctx->err = "fgpu_scan_long_pairs: tai must be a power of two >= 128, n_hash 1..32"; return FGPU_ERR_ARG; }
        // FGPU_DEBUG_LONG_PAIRS_NOMEM=1 (tests): as if the filter's working state -- 4 bytes per bit -- did not fit, so that the hosts' way on can be tested
        static const bool dbg_nomem = getenv("FGPU_DEBUG_LONG_PAIRS_NOMEM") != nullptr;
        hipError_t e = dbg_nomem ? hipErrorOutOfMemory : hipMalloc(&lp.bits, tai / 8);
        if (e == hipSuccess) e = hipMalloc(&lp.first, tai * 4);
        if (e != hipSuccess) {
            if (lp.bits) hipFree(lp.bits);
            lp.bits = nullptr;
            (void)hipGetLastError();
            ctx->err = std::string("fgpu_scan_long_pairs: hipMalloc of the filter and its first-set times (4 bytes per bit) failed: ") + hipGetErrorString(e) +
                       " -- hosts run the loop themselves over fgpu_scan_take_stops' lists then (host/pair_loop.h)";
            return FGPU_ERR_NOMEM;
        }
        lp.tai = tai;
        lp.n_hash = n_hash;
    }
    int rc = fgpu_ensure(ctx, &lp.dev, 256);
    if (rc) return rc;
    if (!lp.flips_host) FGPU_HIP(hipHostMalloc((void**)&lp.flips_host, 64));
    if (!lp.ev_open) FGPU_HIP(hipEventCreateWithFlags(&lp.ev_open, hipEventDisableTiming));
    lp.mode = mode;
    return fgpu_long_pairs_reset(ctx);
}

int fgpu_scan_long_pairs_download(fgpu_ctx* ctx, uint8_t* out, uint64_t n_bytes, uint64_t* empty_count, uint64_t* not_empty_count) {
    if (!ctx) return FGPU_ERR_ARG;
    LongPairs& lp = ctx->lp;
    if (!lp.mode) { ctx->err = "fgpu_scan_long_pairs_download without fgpu_scan_long_pairs"; return FGPU_ERR_STATE; }
    if (ctx->phase != 0) { ctx->err = "fgpu_scan_long_pairs_download while a pass is open"; return FGPU_ERR_STATE; }
    if (out && (lp.mode != FGPU_LONG_PAIRS_FILTER || n_bytes != lp.tai / 8)) { ctx->err = "fgpu_scan_long_pairs_download: the filter has tai / 8 bytes"; return FGPU_ERR_ARG; }
    FGPU_HIP(hipSetDevice(ctx->prm.device));
    if (int rc = fgpu_long_pairs_close(ctx)) return rc;       // the last batch's rounds (fgpu_scan_end has closed it already)
    unsigned long long c[5] = {0, 0, 0, 0, 0};
    FGPU_HIP(hipMemcpyAsync(c, lp.dev.p, sizeof(c), hipMemcpyDeviceToHost, ctx->stream));
    if (out) FGPU_HIP(hipMemcpyAsync(out, lp.bits, n_bytes, hipMemcpyDeviceToHost, ctx->stream));
    FGPU_HIP(fgpu_sync_stream(ctx, ctx->stream));
    if (empty_count) *empty_count = c[3] + lp.empty_host;
    if (not_empty_count) *not_empty_count = c[4];
    return FGPU_OK;
}

int fgpu_diag_long_pairs(fgpu_ctx* ctx, uint64_t out[6]) {
    if (!ctx || !out) return FGPU_ERR_ARG;
    LongPairs& lp = ctx->lp;
    for (int i = 0; i < 6; i++) out[i] = 0;
    if (!lp.mode) return FGPU_OK;
    unsigned long long c[3] = {0, 0, 0};
    FGPU_HIP(hipSetDevice(ctx->prm.device));
    if (int rc = fgpu_long_pairs_close(ctx)) return rc;
    FGPU_HIP(hipMemcpyAsync(c, lp.dev.p, sizeof(c), hipMemcpyDeviceToHost, ctx->stream));   // (behind the batches queued on the context's stream)
    FGPU_HIP(fgpu_sync_stream(ctx, ctx->stream));
    out[0] = c[0] + c[1];   // items: first-end k-mers of read pairs with two non-empty lists
    out[1] = c[1];          // of those, paired against the filter as their batch found it
    out[2] = c[2];          // addPair calls
    out[3] = lp.rounds;     // evaluation rounds, all batches
    out[4] = lp.max_rounds; // most rounds one batch needed
    out[5] = lp.batches;
    return FGPU_OK;
}

}  // extern "C"
