// scan_walk.hip — pass 2, the ORDER-DEPENDENT part: the junction walk, run in parallel and still exact.
//
// Replaces ReadScanner::find_next_junction / scan_forward / add_fake_junction (src/ReadScanner.cpp:61-231)
// and the JunctionMap operations they use (utils/JunctionMap.cpp:533-574, utils/Junction.cpp:59-72).
//
// The reference walks the valid pieces one after another and both reads and writes one global
// unordered_map<kmer, Junction>: a piece sees junctions created (and distances raised) by every earlier
// piece.  Everything a piece reads or writes in that map is keyed by a k-mer that occurs on the piece, and
// everything it can WRITE is keyed by one of its "candidate" positions:
//      positions already in the map, positions flagged by testForJunction, positions the spacer rule can
//      reach, and the middle k-mer where add_fake_junction plants a junction.
// So two pieces can only interact if a candidate k-mer of one occurs (at any position) on the other.
// Per scheduling window of W consecutive pieces:
//   A  k_walk_register candidate positions (in the map as of the batch's snapshot planes nF/nB, flagged, fake-junction or spacer
//                      positions) register their k-mer's hash in a small window table, and so do the keys created since that snapshot
//                      was taken (lists kept by k_delta_collect: the batches walked since, this batch's earlier windows)
//   B  k_walk_link     every position: probe the window table; a hit unions the piece with the candidate's owner
//   C  k_walk_cluster  flatten the union-find; list each cluster's members in ascending piece order
//   D  k_walk          one thread per cluster replays its pieces IN ORDER against the live table; clusters are
//                      disjoint in the keys they touch, so they run concurrently without changing any result
//   D' k_walk_ko       clusters of 64 pieces and more (repeats at high coverage), once a scan has shown one: one piece per wave, the
//                      ACCESSES ordered per junction k-mer by turn counters (k_ko_prepare, k_ko_rank), so the pieces of a cluster
//                      overlap; k_walk leaves those clusters alone.  (k_walk_par, off: the out-of-order walk of clusters whose
//                      pieces change nothing another piece reads -- exact, measured, does not pay.)
//   E  k_delta_collect the keys the window created go to the batch's list; k_walk_reset_uf (side stream) resets the union-find arrays
//                      and the window table's presence filter of this parity (the table itself needs no cleaning: epoch-tagged entries)
// Windows run one after another on the stream, so a later window sees everything earlier ones wrote.
// The result is the reference's map, record for record; creation stamps (global piece number, half-step)
// give the reference's insertion order back for the dump.
//
// Junction table: open addressing on the CANONICAL k-mer; one slot serves both orientations of the key
// (orientation 0: key == canon, orientation 1: key == revcomp(canon)).
//   jkeys[slot]  = canon | present(orient0) << 62 | present(orient1) << 63 ; empty = ~0
//   jrecs[slot][orient] 16 bytes : dist[5] cov[4] linked(bitmask) pad[6]
//   jstamps[slot][orient]         : creation stamp
#include <cstring>
#include <algorithm>

#include "fgpu_ctx.h"
#include "fgpu_flags.h"

#include <rocprim/rocprim.hpp>

namespace {

constexpr uint64_t J_EMPTY = ~0ULL;
constexpr uint64_t J_KEYMASK = (1ULL << 62) - 1;
constexpr uint32_t U_INF = 0xFFFFFFFFu;
constexpr uint64_t J_PROBE_LIMIT = 1ULL << 14;
// creation stamp = (global piece number << STAMP_SHIFT) | half-step of the visit (STAMP_FAKE for add_fake_junction's record, which
// is created after every half-step of its piece): half-steps run to 2 x windows, so a piece may span up to 2^19 - 1 windows
// (fgpu_stage_scan_walk refuses longer reads with FGPU_ERR_CAPACITY); 44 bits are left for the piece number.
constexpr int STAMP_SHIFT = 20;
constexpr uint64_t STAMP_FAKE = (1ULL << STAMP_SHIFT) - 1;

struct JTable {
    uint64_t* keys;
    uint8_t* recs;
    uint64_t* stamps;
    uint64_t mask;         // capacity - 1
    uint32_t* filter;      // presence filter: one bit per hashed canonical k-mer that owns a slot (2 bits per slot of capacity)
    uint64_t filter_mask;  // filter bits - 1
};

// The 32-bit hash of a canonical k-mer that every table of the walk stage works from: h32 = low half of fd_mix(canon).  It is computed
// ONCE per position and batch (k_need_lookup writes the plane `kh`, 4 bytes per position); the per-window kernels read it back
// instead of extracting, reverse-complementing and mixing the k-mer again (which, measured, was NOT what bounded them: they wait on
// dependent loads and on the window table's atomics -- but it is what lets k_walk_register and k_walk_link work without the k-mers).
__device__ __forceinline__ uint32_t jt_h32(uint64_t canon) { return (uint32_t)fd_mix(canon); }
// junction-table slot (capacities up to 2^32) and filter bit (a multiplicative scramble: other bits than the slot's low ones decide)
__device__ __forceinline__ uint64_t jt_filter_bit_h(const JTable& jt, uint32_t h32) {
    return (((uint64_t)(h32 * 0x9E3779B1u) << 16) ^ (uint64_t)(h32 >> 7)) & jt.filter_mask;
}
__device__ __forceinline__ uint64_t jt_filter_bit(const JTable& jt, uint64_t canon) { return jt_filter_bit_h(jt, jt_h32(canon)); }

struct WTable {
    uint64_t* keys;      // epoch << 56 | h32 << 24 | owner (see wt_register)
    uint32_t* bits;      // presence filter of 2^(32 - fshift) bits in front of the table (zeroed by k_walk_reset_uf of the window two before)
    uint64_t mask;
    uint64_t epoch;      // number of the window (1..255) << 56: entries of any other epoch count as empty slots
    uint32_t fshift;     // 32 - log2(bits of the presence filter)
};

// scheduling window = all pieces whose first window lies in [lo, hi); filled by k_walk_setup
struct WinDesc {
    uint32_t first_piece;   // index in the batch's piece list
    uint32_t n;             // pieces in the window
    uint64_t lo, hi;
};

struct Planes {
    const uint64_t* codes;
    const uint64_t* pm;
    const uint64_t* ps;
    const uint32_t* prefix;
    const uint64_t *ff, *fb, *cf0, *cf1, *cb0, *cb1;
    uint64_t *inF, *inB;
    const uint2* pieces;
    uint64_t* lk;
    const uint64_t* need;
    unsigned long long *sF, *sB;   // junction visits (FGPU_FLAG_RECORD_STOPS), else nullptr
    const uint32_t* kh;            // 32-bit hash of the canonical k-mer of every position inside a piece (k_need_lookup)
    unsigned long long* cr;        // positions at which this batch's walk created a junction record (either facing): the delta of later windows
};

__device__ __forceinline__ uint64_t ld_agent(const uint64_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t ld_agent(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- junction table ---------------------------------------------------------------------------
// read-only lookup (snapshot kernels): plain loads
__device__ __forceinline__ uint32_t jt_present_snapshot(const JTable& jt, uint64_t canon) {
    uint64_t s = fd_mix(canon) & jt.mask;
    for (uint64_t n = 0; n <= jt.mask; n++) {
        uint64_t w = jt.keys[s];
        if (w == J_EMPTY) return 0;
        if ((w & J_KEYMASK) == canon) return (uint32_t)(w >> 62);
        s = (s + 1) & jt.mask;
    }
    return 0;
}

// live lookup inside the walk: agent-scope loads (L1 bypass), see DESIGN.md "visibility"
__device__ __forceinline__ bool jt_find_live(const JTable& jt, uint64_t canon, uint64_t& slot, uint32_t& present) {
    uint64_t s = fd_mix(canon) & jt.mask;
    for (uint64_t n = 0; n <= jt.mask; n++) {
        uint64_t w = ld_agent(&jt.keys[s]);
        if (w == J_EMPTY) return false;
        if ((w & J_KEYMASK) == canon) { slot = s; present = (uint32_t)(w >> 62); return true; }
        s = (s + 1) & jt.mask;
    }
    return false;
}

// find or claim the slot of canon; returns false when the table is full.  `w_first` is the (already loaded) key word of the
// home slot, so that the caller can have the record of the home slot in flight at the same time.
__device__ __forceinline__ bool jt_find_or_claim(const JTable& jt, uint64_t canon, uint64_t home, uint64_t w_first, uint64_t& slot,
                                                 uint32_t& present, DevCounters* cnt) {
    uint64_t s = home;
    // The host keeps the table below a quarter full between batches (fgpu_scan_grow); a probe sequence this long means one batch
    // has outgrown it: report "full" now instead of crawling through a saturated table (16 M slots x millions of pieces)
    const uint64_t limit = jt.mask < J_PROBE_LIMIT ? jt.mask : J_PROBE_LIMIT;
    for (uint64_t n = 0; n <= limit; n++) {
        // once some walk has reported the overflow this scan is void: the others stop crawling.  Looked at only on long probe
        // sequences -- a load of that one word from every piece walk queues up in its L2 channel (+8 ms per step, measured)
        if ((n & 255) == 255 && (ld_agent((const uint64_t*)&cnt->error_flags) & 1ULL)) return false;
        uint64_t w = n == 0 ? w_first : ld_agent(&jt.keys[s]);
        if (w == J_EMPTY) {
            unsigned long long old = atomicCAS((unsigned long long*)&jt.keys[s], (unsigned long long)J_EMPTY, (unsigned long long)canon);
            if (old == J_EMPTY) {
                slot = s;
                present = 0;
                return true;
            }
            w = old;
        }
        if ((w & J_KEYMASK) == canon) { slot = s; present = (uint32_t)(w >> 62); return true; }
        s = (s + 1) & jt.mask;
    }
    return false;
}

// ---- union-find over the pieces of one window --------------------------------------------------
__device__ __forceinline__ uint32_t uf_find(uint32_t* parent, uint32_t x) {
    uint32_t p = ld_agent(&parent[x]);
    while (p != x) {
        uint32_t gp = ld_agent(&parent[p]);
        if (gp != p) atomicMin(&parent[x], gp);   // path halving; only ever lowers a pointer towards the root
        x = p;
        p = gp;
    }
    return x;
}

// roots are the smallest index of their set
__device__ __forceinline__ void uf_union(uint32_t* parent, uint32_t a, uint32_t b) {
    for (;;) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a < b) { uint32_t t = a; a = b; b = t; }   // a > b: hook a under b
        uint32_t old = atomicCAS(&parent[a], a, b);
        if (old == a) return;
    }
}

// ---- window table -------------------------------------------------------------------------------
// One 64-bit word per slot: the window's epoch (8 bits) over the key's 32-bit hash over the 24-bit index of the smallest piece
// that registered the key (its owner).  A key that is new in the window -- most are -- costs ONE atomic (the CAS that claims the
// slot carries the owner), a later one a load and an atomicMin.  Two keys with one hash share an entry: their pieces end up in one
// cluster and the positions count as candidates of each other, which only orders more than necessary -- every later use compares
// full k-mers (created_bits) or is a union.
// The epoch makes the clean-up kernel of earlier versions unnecessary (3.8 ms per step plus a launch and two cross-stream events
// per window): an entry written by another window is simply an empty slot -- the table is wiped once every 255 windows.
constexpr int W_OWNER_BITS = 24;
constexpr uint64_t W_OWNER_MASK = (1ULL << W_OWNER_BITS) - 1;
constexpr uint64_t W_EPOCH_MASK = 0xFFULL << 56;
constexpr uint32_t W_NO_OWNER = (uint32_t)W_OWNER_MASK;   // a key registered by k_walk_delta: no piece of the window owns it yet
__device__ __forceinline__ uint32_t wt_filter_bit(const WTable& wt, uint32_t h32) { return (h32 * 0x85EBCA6Bu) >> wt.fshift; }

__device__ __forceinline__ void wt_register(const WTable& wt, uint32_t* parent, uint32_t h32, uint32_t piece, DevCounters* cnt) {
    const unsigned long long mine = (unsigned long long)(wt.epoch | ((uint64_t)h32 << W_OWNER_BITS) | (uint64_t)piece);
    uint64_t s = (uint64_t)h32 & wt.mask;
    for (uint64_t n = 0; n <= wt.mask; n++) {
        unsigned long long v = wt.keys[s];                      // a stale value only costs the CAS / atomicMin below their effect
        if ((v & W_EPOCH_MASK) != wt.epoch) {                   // left by an earlier window: free
            const unsigned long long old = atomicCAS((unsigned long long*)&wt.keys[s], v, mine);
            if (old == v) {
                const uint32_t b = wt_filter_bit(wt, h32);
                atomicOr(&wt.bits[b >> 5], 1u << (b & 31));
                return;
            }
            v = old;                                            // somebody of this window was faster
        }
        if ((v & W_EPOCH_MASK) == wt.epoch && (uint32_t)(v >> W_OWNER_BITS) == h32) {
            if (piece == W_NO_OWNER) return;                    // a delta key that is there already
            const unsigned long long prev = atomicMin((unsigned long long*)&wt.keys[s], mine);
            const uint32_t prev_owner = (uint32_t)(prev & W_OWNER_MASK);
            if (prev_owner != piece && prev_owner != W_NO_OWNER) uf_union(parent, piece, prev_owner);
            return;
        }
        s = (s + 1) & wt.mask;
    }
    atomicOr(&cnt->error_flags, 2ULL);
}

__device__ __forceinline__ uint32_t wt_owner(const WTable& wt, uint32_t h32, uint64_t& slot) {
    const uint32_t b = wt_filter_bit(wt, h32);
    if (!((wt.bits[b >> 5] >> (b & 31)) & 1u)) return U_INF;
    uint64_t s = (uint64_t)h32 & wt.mask;
    for (uint64_t n = 0; n <= wt.mask; n++) {
        const uint64_t w = wt.keys[s];
        if ((w & W_EPOCH_MASK) != wt.epoch) return U_INF;
        if ((uint32_t)(w >> W_OWNER_BITS) == h32) { slot = s; return (uint32_t)(w & W_OWNER_MASK); }
        s = (s + 1) & wt.mask;
    }
    return U_INF;
}

// piece (index in the batch) that contains window position p; requires pm[p] == 1
__device__ __forceinline__ uint32_t piece_of(const Planes& pl, uint64_t p) {
    uint64_t w = pl.ps[p >> 6];
    int o = (int)(p & 63);
    uint64_t upto = o == 63 ? w : (w & ((2ULL << o) - 1));
    return pl.prefix[p >> 6] + (uint32_t)__popcll(upto) - 1;
}

// ---- A: snapshot lookups + candidate registration -------------------------------------------------
// rank of position x among the piece starts = number of ps bits at positions < x
__device__ __forceinline__ uint32_t ps_rank(const Planes& pl, uint64_t x) {
    uint64_t w = pl.ps[x >> 6];
    int o = (int)(x & 63);
    return pl.prefix[x >> 6] + (uint32_t)__popcll(w & ((1ULL << o) - 1));
}

// every kernel of a window derives the window's piece range itself (two rank queries): no set-up launch
__device__ __forceinline__ WinDesc make_window(const Planes& pl, uint64_t lo, uint64_t hi) {
    WinDesc wd;
    wd.first_piece = ps_rank(pl, lo);
    wd.n = ps_rank(pl, hi) - wd.first_piece;
    wd.lo = lo;
    wd.hi = hi;
    return wd;
}

// does window position p (pm[p] == 1) belong to a piece of this window?  returns the window-local piece index
__device__ __forceinline__ bool piece_in_window(const Planes& pl, const WinDesc& wd, uint64_t p, uint32_t& li, uint2& pc) {
    uint32_t pi = piece_of(pl, p);
    if (pi < wd.first_piece || pi >= wd.first_piece + wd.n) return false;
    li = pi - wd.first_piece;
    pc = pl.pieces[pi];
    return true;
}

// In-map bits are NOT looked up per window any more (that was 0.46e9 filter probes and ~5e7 table look-ups per step on the walk's
// critical queue: k_walk_lookup 12.6 ms alone, 24.8 ms beside the pure stage): the planes nF / nB that the pure stage's k_need_lookup
// made for the whole batch are the walk's snapshot.  They are STALE by the time a window is walked -- taken while the previous batch was
// still being walked -- so every key created since then is registered as a candidate as well (k_walk_delta): its positions become lk
// positions, and at lk positions the walk asks the LIVE table (created_bits).  in-map(K) = K in the snapshot, or K created since =
// K registered, looked up live by the one cluster that holds every piece K occurs on: exact.
// ONE launch does all registering of a window; the grid has three parts:
//   blocks [0, word_blocks)              one thread per 64-position word of the window: the few candidate bits of the word (in the map as of
//                                        the snapshot, or testForJunction fired) register; a thread per POSITION -- what the look-up kernel
//                                        of earlier versions was -- spent its time on the dependent loads of 65 k tiny blocks
//   the next piece_blocks                one thread per piece: add_fake_junction's k-mer (ReadScanner.cpp:94) and, in pieces long enough, the
//                                        positions the spacer rule can reach (:72)
//   the last delta_blocks                the delta: keys created since the snapshot planes were made -- the lists k_delta_collect keeps of
//                                        the batches walked since, and of this batch's earlier windows
struct DeltaSrc {   // lists of created-key hashes: the batches walked since the snapshot, and this batch's earlier windows (last entry)
    const uint32_t* list[FGPU_DELTA_RING];
    const unsigned long long* count[FGPU_DELTA_RING];
};

__global__ void __launch_bounds__(256) k_walk_register(Planes pl, FdParams fp, WTable wt, uint32_t* parent, uint64_t lo, uint64_t hi,
                                                       uint64_t pos_end, DevCounters* cnt, unsigned word_blocks, unsigned piece_blocks, DeltaSrc ds) {
    const WinDesc wd = make_window(pl, lo, hi);
    if (blockIdx.x < word_blocks) {
        const uint64_t w = (wd.lo >> 6) + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (w * 64 >= pos_end) return;
        unsigned long long m = pl.pm[w] & (pl.inF[w] | pl.inB[w] | pl.ff[w] | pl.fb[w]);
        if (w == (wd.lo >> 6)) m &= ~0ULL << (wd.lo & 63);
        if ((w + 1) * 64 > pos_end) m &= (1ULL << (pos_end & 63)) - 1;
        while (m) {
            const int b = __builtin_ctzll(m);
            m &= m - 1;
            const uint64_t p = w * 64 + (uint64_t)b;
            uint32_t li;
            uint2 pc;
            if (piece_in_window(pl, wd, p, li, pc)) wt_register(wt, parent, pl.kh[p], li, cnt);
        }
    } else if (blockIdx.x < word_blocks + piece_blocks) {
        const uint32_t li = (blockIdx.x - word_blocks) * blockDim.x + threadIdx.x;
        if (li >= wd.n) return;
        const uint2 pc = pl.pieces[wd.first_piece + li];
        const uint32_t len = pc.y + (uint32_t)fp.k - 1;
        wt_register(wt, parent, pl.kh[pc.x + (len / 2 - (uint32_t)fp.k / 2)], li, cnt);
        for (uint32_t q = (uint32_t)fp.max_spacer - 1; q < pc.y; q++) wt_register(wt, parent, pl.kh[pc.x + q], li, cnt);   // 2q + 1 >= 2 spacer - 1
    } else {
        const uint64_t t = (uint64_t)(blockIdx.x - word_blocks - piece_blocks) * blockDim.x + threadIdx.x;
        const uint64_t stride = (uint64_t)(gridDim.x - word_blocks - piece_blocks) * blockDim.x;
        for (int r = 0; r < FGPU_DELTA_RING; r++) {
            if (!ds.list[r]) continue;
            const uint64_t n = *ds.count[r];
            for (uint64_t i = t; i < n; i += stride) wt_register(wt, nullptr, ds.list[r][i], W_NO_OWNER, cnt);
        }
    }
}

// ---- B: link every piece to the owners of the candidate k-mers that occur on it ---------------------
__global__ void __launch_bounds__(256) k_walk_link(Planes pl, FdParams fp, WTable wt, uint32_t* parent, uint64_t lo, uint64_t hi,
                                                   uint64_t pos_end, uint32_t* roots_state) {
    // a fixed grid strides over the window, 1024 positions per block and round: the loads of four positions per thread (hash, then
    // filter word) are in flight together -- one position per thread left this kernel waiting on two dependent loads per 256-thread block
    const WinDesc wd = make_window(pl, lo, hi);
    if (roots_state && blockIdx.x == 0 && threadIdx.x == 0) { roots_state[0] = 0; roots_state[1] = 0; }   // roots listed / handed out (k_walk_cluster, k_walk)
    const uint64_t base0 = wd.lo & ~63ULL;
    constexpr int U = 4;
    for (uint64_t base = base0 + (uint64_t)blockIdx.x * (256 * U); base < pos_end; base += (uint64_t)gridDim.x * (256 * U)) {
        uint64_t p[U];
        uint32_t h[U], fw[U];
        bool act[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            p[u] = base + (uint64_t)u * 256 + threadIdx.x;
            act[u] = p[u] >= wd.lo && p[u] < pos_end && ((pl.pm[p[u] >> 6] >> (p[u] & 63)) & 1ULL);
        }
#pragma unroll
        for (int u = 0; u < U; u++) h[u] = act[u] ? pl.kh[p[u]] : 0u;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t b = wt_filter_bit(wt, h[u]);
            fw[u] = act[u] ? ((wt.bits[b >> 5] >> (b & 31)) & 1u) : 0u;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            bool hit = false;
            if (fw[u]) {
                uint64_t slot = 0;
                uint32_t owner = wt_owner(wt, h[u], slot);
                if (owner != U_INF) {
                    uint32_t li;
                    uint2 pc;
                    if (piece_in_window(pl, wd, p[u], li, pc)) {
                        hit = true;
                        if (owner == W_NO_OWNER) {   // a key created since the snapshot: the first piece it occurs on becomes its owner, the others join it
                            const unsigned long long mine = (unsigned long long)(wt.epoch | ((uint64_t)h[u] << W_OWNER_BITS) | (uint64_t)li);
                            owner = (uint32_t)(atomicMin((unsigned long long*)&wt.keys[slot], mine) & W_OWNER_MASK);
                        }
                        if (owner != li && owner != W_NO_OWNER) uf_union(parent, li, owner);
                    }
                }
            }
            // lk plane: the positions that hold a registered k-mer -- the only ones at which the walk has to ask the live table
            const uint64_t m = __ballot(hit);
            if (fd_lane() == 0 && p[u] < ((pos_end + 63) & ~63ULL)) pl.lk[p[u] >> 6] = m;
        }
    }
}

// ---- B': the link pass of a PREPARED batch of a read shard, by a candidate plane made off the chain (round 6) --------------------------
// On the chain of hand-overs a prepared batch pays two per-position passes over its hash plane: k_refresh_lookup (the keys created since the
// preview, merged into the in-map planes through a small filter) and k_walk_link (every position against the window table).  The second one
// asks a question whose answer is almost entirely known BEFORE the table arrives: the keys a window registers are
//   (a) candidates by the planes as the preview left them (in the map, flagged), by the pieces' shapes (fake junction, spacer rule): static;
//   (b) positions whose in-map bit the merge has just set: their k-mer is in the filter of new keys;
//   (c) keys this batch's own earlier windows created (the delta list): only for windows with lo > 0.
// So while the rank waits (fgpu_scan_refresh_prepared) every prepared batch gets a plane `cand`: positions whose hash is in a filter of all
// hashes of kind (a) -- a superset of the positions k_walk_link would find for them, whatever the window --, the merge ORs in the positions
// that hit the filter of new keys (a superset for (b): every position with the hash of such a key hits it too), and the link pass of a
// window with lo == 0 visits the set bits only.  Windows with lo > 0 are linked in full.  FGPU_DEBUG_DELTA_CHECK=1 runs the full pass behind
// the sparse one and compares the lk planes word by word.
constexpr int CAND_FILTER_LOG2 = 27;
__device__ __forceinline__ uint32_t cand_bit(uint32_t h32) { return (h32 * 0xC2B2AE35u) >> (32 - CAND_FILTER_LOG2); }

__global__ void __launch_bounds__(256) k_cand_mark(const uint64_t* __restrict__ pm, const uint64_t* __restrict__ inF, const uint64_t* __restrict__ inB,
                                                   const uint64_t* __restrict__ ff, const uint64_t* __restrict__ fb, const uint32_t* __restrict__ kh,
                                                   uint64_t n_words, const uint2* __restrict__ pieces, uint64_t n_pieces, FdParams fp,
                                                   uint32_t* __restrict__ filt, unsigned word_blocks) {
    if (blockIdx.x < word_blocks) {
        for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (uint64_t)word_blocks * blockDim.x) {
            unsigned long long m = pm[w] & (inF[w] | inB[w] | ff[w] | fb[w]);
            while (m) {
                const int b = __builtin_ctzll(m);
                m &= m - 1;
                const uint32_t cb = cand_bit(kh[w * 64 + (uint64_t)b]);
                atomicOr(&filt[cb >> 5], 1u << (cb & 31));
            }
        }
    } else {   // the pieces' own candidates, as k_walk_register takes them
        const uint64_t stride = (uint64_t)(gridDim.x - word_blocks) * blockDim.x;
        for (uint64_t i = (uint64_t)(blockIdx.x - word_blocks) * blockDim.x + threadIdx.x; i < n_pieces; i += stride) {
            const uint2 pc = pieces[i];
            const uint32_t len = pc.y + (uint32_t)fp.k - 1;
            uint32_t cb = cand_bit(kh[pc.x + (len / 2 - (uint32_t)fp.k / 2)]);
            atomicOr(&filt[cb >> 5], 1u << (cb & 31));
            for (uint32_t q = (uint32_t)fp.max_spacer - 1; q < pc.y; q++) {
                cb = cand_bit(kh[pc.x + q]);
                atomicOr(&filt[cb >> 5], 1u << (cb & 31));
            }
        }
    }
}

__global__ void __launch_bounds__(256) k_cand_probe(const uint64_t* __restrict__ pm, const uint32_t* __restrict__ kh, uint64_t n_words,
                                                    const uint32_t* __restrict__ filt, uint64_t* __restrict__ cand) {
    const uint64_t total = n_words * 64;
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (uint64_t)gridDim.x * blockDim.x) {
        bool hit = false;
        if ((pm[p >> 6] >> (p & 63)) & 1ULL) {
            const uint32_t cb = cand_bit(kh[p]);
            hit = (filt[cb >> 5] >> (cb & 31)) & 1u;
        }
        const uint64_t m = __ballot(hit);
        if (fd_lane() == 0) cand[p >> 6] = m;
    }
}

__global__ void __launch_bounds__(256) k_walk_link_sparse(Planes pl, FdParams fp, WTable wt, uint32_t* parent, uint64_t lo, uint64_t hi,
                                                          uint64_t pos_end, uint32_t* roots_state, const uint64_t* __restrict__ cand, uint64_t* lk_out) {
    const WinDesc wd = make_window(pl, lo, hi);
    if (roots_state && blockIdx.x == 0 && threadIdx.x == 0) { roots_state[0] = 0; roots_state[1] = 0; }
    const uint64_t w0 = wd.lo >> 6, w1 = (pos_end + 63) >> 6;
    for (uint64_t w = w0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < w1; w += (uint64_t)gridDim.x * blockDim.x) {
        unsigned long long m = cand[w] & pl.pm[w];
        if (w == w0) m &= ~0ULL << (wd.lo & 63);
        if ((w + 1) * 64 > pos_end) m &= (1ULL << (pos_end & 63)) - 1;
        unsigned long long out = 0;
        while (m) {
            const int b = __builtin_ctzll(m);
            m &= m - 1;
            const uint64_t p = w * 64 + (uint64_t)b;
            const uint32_t h = pl.kh[p];
            uint64_t slot = 0;
            uint32_t owner = wt_owner(wt, h, slot);
            if (owner == U_INF) continue;
            uint32_t li;
            uint2 pc;
            if (!piece_in_window(pl, wd, p, li, pc)) continue;
            out |= 1ULL << b;
            if (owner == W_NO_OWNER) {
                const unsigned long long mine = (unsigned long long)(wt.epoch | ((uint64_t)h << W_OWNER_BITS) | (uint64_t)li);
                owner = (uint32_t)(atomicMin((unsigned long long*)&wt.keys[slot], mine) & W_OWNER_MASK);
            }
            if (owner != li && owner != W_NO_OWNER) uf_union(parent, li, owner);
        }
        lk_out[w] = out;
    }
}

// ---- C: clusters -> member lists ---------------------------------------------------------------------
// One pass over the pieces of the window (the union-find is final: every union happened in the kernels before):
// flat root of every piece, and every follower pushes itself onto its root's singly linked list.  No scan, no
// second launch; the leader of a cluster sorts its (short) list when it walks.
// weight / heavy_w (key-ordered walk by WEIGHT, see ko_cluster): the lk positions of a cluster's pieces, summed per root.
__global__ void __launch_bounds__(256) k_walk_cluster(const uint32_t* __restrict__ parent, uint32_t* count, uint32_t* head,
                                                      uint32_t* __restrict__ flat, uint32_t* __restrict__ next, Planes pl, uint64_t lo,
                                                      uint64_t hi, WinDesc* wd_out, DevCounters* cnt, uint32_t* weight, uint32_t heavy_w,
                                                      uint32_t* __restrict__ root_list, uint32_t* roots_state) {
    const WinDesc wd = make_window(pl, lo, hi);
    const uint32_t n = wd.n;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *wd_out = wd;                  // the walk kernel reads it with one uniform load
        cnt->pad2 = 0;                 // pool cursor of the giant-cluster lists of the walk kernel that follows
    }
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint32_t r = i;
        for (;;) { uint32_t pr = parent[r]; if (pr == r) break; r = pr; }
        flat[i] = r;
        if (r != i) {
            next[i] = atomicExch(&head[r], i);
            atomicAdd(&count[r], 1u);
        }
        {   // the clusters' leaders, as a list k_walk hands out (one reservation per wave)
            const bool is_root = r == i;
            const unsigned long long m = __ballot(is_root);
            if (m) {
                const int first = __builtin_ctzll(m);
                uint32_t base = 0;
                if (fd_lane() == first) base = atomicAdd(&roots_state[0], (uint32_t)__popcll(m));
                base = (uint32_t)__shfl((int)base, first, 64);
                if (is_root) root_list[base + (uint32_t)__popcll(m & ((1ULL << fd_lane()) - 1))] = i;
            }
        }
        if (heavy_w) {
            const uint2 pc = pl.pieces[wd.first_piece + i];
            uint32_t w = 0;
            for (uint32_t c = 0; c < 8 && c * 64 < pc.y; c++) {
                const uint32_t rem = pc.y - c * 64;
                w += (uint32_t)__popcll(fd_bits_at(pl.lk, pc.x + 64 * c) & (rem >= 64 ? ~0ULL : ((1ULL << rem) - 1)));
            }
            if (w) atomicAdd(&weight[r], w);
        }
    }
}

// Which clusters the key-ordered walk takes: those of at least `heavy` pieces, and -- heavy_w != 0: callers that expect repeats
// (FGPU_FLAG_KEY_ORDER_FROM_START) -- those whose pieces hold at least heavy_w lk positions between them.  A piece inside a repeat at high
// coverage makes a junction visit at nearly every position: a dozen such pieces on their cluster's one thread (k_walk) are a longer chain
// than seventy ordinary ones, and the longest chain is what a window's k_walk takes (BASELINE config 3's shape through the CLI: k_walk
// 237 ms of pass 2 beside k_walk_ko's 238 ms, all of it clusters of fewer than 32 repeat pieces).
// (measured, FGPU_WALK_KO_WEIGHT: on config 3's shape k_walk + k_walk_ko take 381 / 293 / 249 ms at 256 / 128 / 48 positions; on config 2,
// where a piece holds 2-5 of them, the walk stage grows from 42.9 ms to 45.0 / 63.4 / 80.1 ms at 128 / 64 / 48.  Hence two bars: heavy_w
// positions in all, or three eighths of that where the pieces hold a dozen each -- which ordinary pieces never do.)
// heavy_w packs the rule by weight: bits 0-15 the bar for a cluster's lk positions in all, bits 16-23 the positions per piece that make its
// pieces "repeat pieces", bits 24-31 the bar in all for clusters of such pieces (fgpu_stage_scan_walk; 0 = rule off)
__device__ __forceinline__ bool ko_cluster(uint32_t followers, uint32_t weight, uint32_t heavy, uint32_t heavy_w) {
    const uint32_t total = heavy_w & 0xFFFFu, per_piece = (heavy_w >> 16) & 0xFFu, total_rep = heavy_w >> 24;
    return followers + 1 >= heavy || (heavy_w && (weight >= total || (per_piece && weight >= total_rep && weight >= per_piece * (followers + 1))));
}

// ---- D: the walk ---------------------------------------------------------------------------------------
// Everything the walk needs to know about the first 128 windows of a piece sits in registers: the eight bit
// planes (2 words each) are fetched with one burst of independent loads, after which finding the next event,
// classifying it and summing NbJCheckKmer increments is bit arithmetic.  Windows beyond 128 (reads longer than
// ~150 bp) go through the same code with plane words fetched on demand.
struct PieceView {
    uint64_t p0;
    uint32_t nwin;
    // two words per plane, as SCALAR members: runtime-indexed arrays would be demoted to scratch memory
    uint64_t inF0, inF1, inB0, inB1, fF0, fF1, fB0, fB1, c0F0, c0F1, c1F0, c1F1, c0B0, c0B1, c1B0, c1B1;
    uint64_t xF0, xF1, xB0, xB1;   // positions whose key this thread's cluster created during the current window
    uint64_t lk0, lk1;             // positions holding a registered candidate k-mer (where created keys can sit)
    uint64_t nd0, nd1;             // positions whose flags the pure stage evaluated (need plane)
    uint64_t cbase;                // stream position of the first base held in cw0 (multiple of 32)
};

// ---- the key-ordered walk of large clusters (k_walk_ko) ------------------------------------------------------------------------------------
// A cluster's pieces are walked in file order because of what they read and write in the junction map -- and they read and write it one
// k-mer at a time.  Ordering the ACCESSES per k-mer instead of the pieces per cluster keeps every read and write where the sequential run
// has it and lets the pieces of a cluster overlap: piece i+1 follows piece i through the k-mers they share, a few positions behind, instead
// of starting when piece i has ended.  Every occurrence of a registered k-mer (an lk position) on a piece of a large cluster gets a rank
// among the occurrences of that k-mer, by (piece, position); a k-mer has a turn counter; a piece may look its k-mer up (and visit, create,
// update) when the counter equals its occurrence's rank, and sets it to rank + 1 when it has passed the position and stored what it changed.
// Positions the walk skips are passed just the same (their turn is taken and given back), so the counter of a k-mer advances in piece order
// whatever the pieces do there.  A piece waits only for pieces before it, which are running or done (pieces are handed out in file order from
// a ticket counter): no cycle.  One piece per wave (lane 0), so a waiting piece never holds back a lane it waits for.
struct KoTables {
    uint32_t* hk_key;     // heavy-key table: 32-bit hash of the canonical k-mer (the window table's notion of a key), KO_EMPTY = free
    uint32_t* hk_head;    // head of the k-mer's occurrence list (k_ko_prepare), U_INF = none
    uint32_t* hk_turn;    // the turn counter
    uint32_t hk_mask;
    uint32_t* occ_entry;  // per occurrence: its k-mer's table entry
    uint32_t* occ_rank;   // per occurrence: its rank among the k-mer's occurrences
    uint32_t* occ_next;   // list link
    uint64_t* occ_id;     // piece << 32 | position: the order of the occurrences of a k-mer
    uint32_t occ_cap;
    uint32_t* piece_base; // per piece of the window: first occurrence number (its lk positions in ascending order follow)
    uint32_t* state;      // [0] occurrences reserved, [1] bit 0: a table overflowed (every cluster is walked in order), [2] ticket of k_walk_ko
    uint32_t* bad;        // per root: the cluster holds a piece the key-ordered walk does not take (more than 128 windows)
    unsigned long long* trace;   // -DFGPU_KO_TRACE (measurement build): [0] records written; 4 words per walked piece from [4] on, else nullptr
};
constexpr uint32_t KO_EMPTY = 0xFFFFFFFFu;
#ifndef FGPU_KO_WAIT_S
#define FGPU_KO_WAIT_S 60
#endif
constexpr unsigned long long KO_WAIT_LIMIT_TICKS = (unsigned long long)FGPU_KO_WAIT_S * 100000000ULL;   // 60 s at one turn counter: give up loudly (error bit 8), never hang

// A piece's lk positions and the positions looked up and found absent (per facing: not events after all), 64 windows per word, in LDS: the
// one walking lane of the block indexes them by chunk at run time (registers picked by an index end up in scratch memory, see pv_codes).
constexpr uint32_t KO_CHUNKS = 8;              // pieces of up to 512 windows; longer ones keep their cluster with k_walk
__device__ __forceinline__ uint64_t* ko_masks() {
    __shared__ uint64_t s_ko[3 * KO_CHUNKS];
    return s_ko;
}
__device__ __forceinline__ uint64_t& ko_lk(uint32_t c) { return ko_masks()[c]; }
__device__ __forceinline__ uint64_t& ko_absent(bool fwd, uint32_t c) { return ko_masks()[(fwd ? 1 : 2) * KO_CHUNKS + c]; }

struct KoHold {   // a k-mer whose turn this piece holds: entry, rank of its first occurrence here, occurrences merged into the hold
    uint32_t e, r, n;
};
struct KoState {
    KoTables kt;
    DevCounters* cnt;       // (timing build)
    uint32_t base;          // first occurrence number of this piece
    uint32_t done;          // every lk position below `done` has been accounted for (passed or held)
    uint32_t mid;           // the position add_fake_junction would use
    KoHold last, cur, fake; // held: the last junction's k-mer (until its record is stored), the position under the cursor, the fake candidate
    uint32_t cur_q;         // the position under the cursor
    int cur_in;             // which hold carries the cursor's k-mer: 0 cur, 1 last, 2 fake (the same k-mer twice on one piece shares a hold)
    unsigned long long wait_acc;       // -DFGPU_KO_TRACE: ticks (10 ns) this piece has spent waiting for turns
    unsigned long long* stamp_base;    // -DFGPU_KO_TRACE: where this piece leaves its per-step time stamps (nullptr: it does not)
    unsigned long long stamp_t0;
    uint32_t stamp_n;
};

struct WalkCtx {
    Planes pl;
    FdParams fp;
    JTable jt;
    DevCounters* cnt;
    const uint32_t* bloom;   // bloo2, for the junction tests the preview did not order (walk_fill_flags)
    // per-thread accumulators
    unsigned long long nb_processed, nb_skipped, nb_jcheck, nb_no_juncs, n_created, n_filled;
    bool created_now;   // set by junction_get
    KoState ko;         // WALK_KO: the piece's turn bookkeeping (by value: through a pointer it lived in scratch memory)
    uint32_t win_seq;   // number of the window being walked (late junction tests are noted with it)
    int fail;           // WALK_PROBE: why this piece cannot be walked out of order (see k_walk_par): 1 would create, 2 would raise a distance, 3 untested positions
    int dbg;            // FGPU_DEBUG_WALK bits (timing experiments only; results are wrong when non-zero)
};

// How a piece is walked.  WALK_SEQ: in its cluster's order by the cluster's one thread, reading and writing the junction table (the
// reference's semantics as they stand).  WALK_PROBE / WALK_COMMIT: the two halves of the out-of-order walk of a large cluster, one thread
// per piece (k_walk_par): PROBE walks read-only and notes whether the piece would change anything a later piece's path can depend on,
// COMMIT walks the same path again and applies what is left -- coverage counts and link flags, both order-free -- with atomics.
enum { WALK_SEQ = 0, WALK_PROBE = 1, WALK_COMMIT = 2, WALK_KO = 3, WALK_OVW = 4 };
#ifdef FGPU_KO_TIMING
#define KO_T0() const unsigned long long ko_t0__ = wall_clock64()
#define KO_T1(cnt, i) atomicAdd(&(cnt)->ko_time[i], wall_clock64() - ko_t0__)
#else
#define KO_T0() do {} while (0)
#define KO_T1(cnt, i) do {} while (0)
#endif


__device__ __forceinline__ uint64_t chunk_mask(uint32_t nwin, uint32_t c) {
    uint32_t base = c * 64;
    if (base >= nwin) return 0;
    uint32_t rem = nwin - base;
    return rem >= 64 ? ~0ULL : ((1ULL << rem) - 1);
}

// plane word for windows [64c, 64c+64) of the piece
__device__ __forceinline__ uint64_t pv_word(const PieceView& v, uint64_t r0, uint64_t r1, const uint64_t* plane, uint32_t c) {
    if (c == 0) return r0;
    if (c == 1) return r1;
    // beyond the 128 windows held in registers: agent-scope loads, because the walk may have patched these words itself
    // (walk_fill_flags) and a plain load could be served from a stale L1 line
    const uint64_t p = v.p0 + 64ULL * c;
    const int o = (int)(p & 63);
    const uint64_t lo = ld_agent(&plane[p >> 6]), hi = ld_agent(&plane[(p >> 6) + 1]);
    return ((lo >> o) | ((hi << 1) << (63 - o))) & chunk_mask(v.nwin, c);
}

// The 192 bases of 2-bit codes around a piece (a whole <= 160-base piece) live in LDS, one column of six words per lane: a k-mer comes out
// of two LDS reads at a computed row.  (As six register members picked by a chain of selects the compiler turned every pick into a 48-byte
// array in SCRATCH memory -- three stores and an indexed load through the vector memory path per k-mer, on the path of every junction visit.)
// The walk kernels run blocks of one wave.
__device__ __forceinline__ uint64_t* pv_codes() {
    __shared__ uint64_t s_codes[6 * 64];
    return s_codes + (threadIdx.x & 63);
}

__device__ __forceinline__ void pv_load(PieceView& v, const Planes& pl, uint64_t p0, uint32_t nwin) {
    v.p0 = p0;
    v.nwin = nwin;
    const uint64_t m0 = chunk_mask(nwin, 0), m1 = chunk_mask(nwin, 1);
    const uint64_t p1 = p0 + 64;
    // one burst of independent loads (a piece's second word is skipped when the piece has at most 64 windows)
    v.inF0 = fd_bits_at(pl.inF, p0) & m0;   v.inF1 = fd_bits_at(pl.inF, p1) & m1;
    v.inB0 = fd_bits_at(pl.inB, p0) & m0;   v.inB1 = fd_bits_at(pl.inB, p1) & m1;
    v.fF0 = fd_bits_at(pl.ff, p0) & m0;     v.fF1 = fd_bits_at(pl.ff, p1) & m1;
    v.fB0 = fd_bits_at(pl.fb, p0) & m0;     v.fB1 = fd_bits_at(pl.fb, p1) & m1;
    v.c0F0 = fd_bits_at(pl.cf0, p0) & m0;   v.c0F1 = fd_bits_at(pl.cf0, p1) & m1;
    v.c1F0 = fd_bits_at(pl.cf1, p0) & m0;   v.c1F1 = fd_bits_at(pl.cf1, p1) & m1;
    v.c0B0 = fd_bits_at(pl.cb0, p0) & m0;   v.c0B1 = fd_bits_at(pl.cb0, p1) & m1;
    v.c1B0 = fd_bits_at(pl.cb1, p0) & m0;   v.c1B1 = fd_bits_at(pl.cb1, p1) & m1;
    v.xF0 = v.xF1 = v.xB0 = v.xB1 = 0;
    v.lk0 = fd_bits_at(pl.lk, p0) & m0;     v.lk1 = fd_bits_at(pl.lk, p1) & m1;
    v.nd0 = fd_bits_at(pl.need, p0) & m0;   v.nd1 = fd_bits_at(pl.need, p1) & m1;
    v.cbase = p0 & ~31ULL;
    const uint64_t* cw = pl.codes + (v.cbase >> 5);   // padded: reading 6 words from any piece start stays inside the buffer
    uint64_t* l = pv_codes();
    l[0 * 64] = cw[0]; l[1 * 64] = cw[1]; l[2 * 64] = cw[2]; l[3 * 64] = cw[3]; l[4 * 64] = cw[4]; l[5 * 64] = cw[5];
}

__device__ __forceinline__ uint64_t pv_cw(const PieceView& v, uint32_t w) {
    (void)v;
    return pv_codes()[w * 64];
}
// k-mer / base at stream position p: from the register copy when it covers p, else from memory
__device__ __forceinline__ uint64_t pv_kmer(const PieceView& v, const uint64_t* codes, uint64_t p, int k) {
    const uint64_t rel = p - v.cbase;
    if (rel + (uint64_t)k > 160) return fd_kmer_at(codes, p, k);
    const uint32_t w = (uint32_t)(rel >> 5);
    const int o = (int)(rel & 31) * 2;
    const uint64_t hi = pv_cw(v, w), lo = pv_cw(v, w + 1);
    return ((hi << o) | ((lo >> 1) >> (63 - o))) >> (64 - 2 * k);
}
__device__ __forceinline__ int pv_base(const PieceView& v, const uint64_t* codes, uint64_t p) {
    const uint64_t rel = p - v.cbase;   // p >= p0 - 1; p0 - 1 can precede cbase only when p0 is a multiple of 32
    if (p < v.cbase || rel >= 192) return fd_base_at(codes, p);
    return (int)((pv_cw(v, (uint32_t)(rel >> 5)) >> (62 - 2 * (int)(rel & 31))) & 3);
}

// In-map bits that the batch's snapshot planes cannot know (keys created since they were made: by earlier batches' walks still in flight
// then, by earlier windows, by earlier pieces of this cluster, by the piece itself): the LIVE table is asked, and only at the piece's lk
// positions -- a key created since the snapshot is a registered candidate of this window (k_walk_register's delta), so its positions are
// lk positions, and every piece it occurs on is in this thread's cluster: what the look-up sees is what the sequential run has.
// Out of line and fed by value on purpose: the rare path stays out of the walk's register allocation.
// (the table as two scalars, not the 48-byte JTable: a struct handed to an out-of-line function by value travels through SCRATCH memory --
// three 16-byte stores before every call and as many loads behind it, on the path of nearly every piece of k_walk)
__device__ __noinline__ uint4 live_bits_impl(const uint64_t* __restrict__ codes, int k, uint64_t p, uint64_t where, uint64_t* jkeys, uint64_t jmask) {
    JTable jt;
    jt.keys = jkeys;
    jt.mask = jmask;
    jt.recs = nullptr; jt.stamps = nullptr; jt.filter = nullptr; jt.filter_mask = 0;
    uint64_t mF = 0, mB = 0;
    // Four candidate positions at a time, their loads in flight TOGETHER: first the four k-mers, then the key words of their four home slots.
    // One position after the other this was two dependent round trips per position -- a piece holds 2-5 such positions, i.e. 4-10 of the 8-10
    // round trips that make a piece's place in its cluster's chain (round 4; k_walk is the longest chain of a window, not throughput).
    while (where) {
        uint32_t idx[4];
        bool on[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            on[j] = where != 0;
            idx[j] = on[j] ? (uint32_t)__builtin_ctzll(where) : 0u;
            where &= where - 1;                          // (0 stays 0)
        }
        uint64_t km[4], rc[4], canon[4], home[4], w[4];
#pragma unroll
        for (int j = 0; j < 4; j++) km[j] = fd_kmer_at(codes, p + idx[j], k);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            rc[j] = fd_revcomp(km[j], k);
            canon[j] = km[j] < rc[j] ? km[j] : rc[j];
            home[j] = fd_mix(canon[j]) & jt.mask;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) w[j] = on[j] ? ld_agent(&jt.keys[home[j]]) : J_EMPTY;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (!on[j] || w[j] == J_EMPTY) continue;
            uint32_t present = 0;
            bool found = (w[j] & J_KEYMASK) == canon[j];
            if (found) present = (uint32_t)(w[j] >> 62);
            else {                                       // not in its home slot: the rest of the probe sequence (rare at the table's load factor)
                uint64_t slot;
                found = jt_find_live(jt, canon[j], slot, present);
            }
            if (found) {
                if ((present >> (km[j] == canon[j] ? 0 : 1)) & 1u) mF |= 1ULL << idx[j];   // forward-facing key = the k-mer itself
                if ((present >> (rc[j] == canon[j] ? 0 : 1)) & 1u) mB |= 1ULL << idx[j];   // backward-facing key = its reverse complement
            }
        }
    }
    return make_uint4((uint32_t)mF, (uint32_t)(mF >> 32), (uint32_t)mB, (uint32_t)(mB >> 32));
}

__device__ __forceinline__ void created_bits(const WalkCtx& wc, const PieceView& v, uint32_t c, uint64_t& mF, uint64_t& mB) {
    if (wc.dbg & 1) { mF = mB = 0; return; }
    const uint32_t base = c * 64;
    // (junctions are never removed during a scan: where the snapshot already has both facings there is nothing the live table could add --
    // inside a repeat at high coverage that is nearly every position, and the look-ups of a piece were a quarter of its walk)
    const uint64_t where = pv_word(v, v.lk0, v.lk1, wc.pl.lk, c) & ~(pv_word(v, v.inF0, v.inF1, wc.pl.inF, c) & pv_word(v, v.inB0, v.inB1, wc.pl.inB, c));
    if (!where) { mF = mB = 0; return; }
    uint4 r = live_bits_impl(wc.pl.codes, wc.fp.k, v.p0 + base, where, wc.jt.keys, wc.jt.mask);
    mF = (uint64_t)r.x | ((uint64_t)r.y << 32);
    mB = (uint64_t)r.z | ((uint64_t)r.w << 32);
}

template <int MODE>
__device__ __forceinline__ void in_map_words(const WalkCtx& wc, const PieceView& v, uint32_t c, uint64_t& mF, uint64_t& mB) {
    if (MODE == WALK_KO) {   // what the map holds is asked when the k-mer's turn has come: until then every registered position may be in it
        mF = c < KO_CHUNKS ? (ko_lk(c) & ~ko_absent(true, c)) : 0ULL;
        mB = c < KO_CHUNKS ? (ko_lk(c) & ~ko_absent(false, c)) : 0ULL;
        return;
    }
    mF = pv_word(v, v.inF0, v.inF1, wc.pl.inF, c);
    mB = pv_word(v, v.inB0, v.inB1, wc.pl.inB, c);
    if (c == 0) { mF |= v.xF0; mB |= v.xB0; return; }
    if (c == 1) { mF |= v.xF1; mB |= v.xB1; return; }
    uint64_t a, b;
    created_bits(wc, v, c, a, b);
    mF |= a;
    mB |= b;
}

// number of set bits of the plane at windows [qa, qb) of the piece
__device__ __forceinline__ uint32_t pv_popc(const PieceView& v, uint64_t r0, uint64_t r1, const uint64_t* plane, uint32_t qa, uint32_t qb) {
    uint32_t s = 0;
    if (qb > v.nwin) qb = v.nwin;
    while (qa < qb) {
        const uint32_t c = qa >> 6;
        uint64_t w = pv_word(v, r0, r1, plane, c) >> (qa & 63);
        const uint32_t n = min(qb - qa, 64u - (qa & 63));
        if (n < 64) w &= (1ULL << n) - 1;
        s += (uint32_t)__popcll(w);
        qa += n;
    }
    return s;
}

// NbJCheckKmer increments of the half-steps t in [t0, t1): backward-facing half-steps are 2q, forward-facing 2q+1
__device__ __forceinline__ uint32_t jcheck_sum(const WalkCtx& wc, const PieceView& v, int t0, int t1) {
    if (t1 <= t0 || (wc.dbg & 4)) return 0;
    const uint32_t bq0 = (uint32_t)((t0 + 1) >> 1), bq1 = (uint32_t)((t1 + 1) >> 1);
    const uint32_t fq0 = (uint32_t)(t0 >> 1), fq1 = (uint32_t)(t1 >> 1);
    return pv_popc(v, v.c0B0, v.c0B1, wc.pl.cb0, bq0, bq1) + 2 * pv_popc(v, v.c1B0, v.c1B1, wc.pl.cb1, bq0, bq1) +
           pv_popc(v, v.c0F0, v.c0F1, wc.pl.cf0, fq0, fq1) + 2 * pv_popc(v, v.c1F0, v.c1F1, wc.pl.cf1, fq0, fq1);
}

// The preview of the pure stage (need plane) says where the walk may stop skipping; only there are testForJunction's
// answers in the flag planes.  The preview is built on a junction map that keeps changing -- distances are also raised by
// reads that travel the other way and link a junction to a farther one -- so now and then the walk scans a window the
// preview left out.  Its junction tests are then evaluated right here (the same code the pure stage runs), patched into
// the planes this walk reads, and the search that needed them is repeated: the result is exact whatever the preview said.
__device__ __noinline__ uint32_t walk_fill_flags(const uint64_t* __restrict__ codes, const uint32_t* __restrict__ bloom, const FdParams fp,
                                                 uint64_t pos, bool has_next, bool has_prev) {
    const uint64_t km = fd_kmer_at(codes, pos, fp.k);
    bool f_f = false, f_b = false;
    int c_f = 0, c_b = 0;
    if (has_next) test_for_junction(km, fd_base_at(codes, pos + fp.k), fp, bloom, f_f, c_f);                   // a window follows: facing forward
    if (has_prev) test_for_junction(fd_revcomp(km, fp.k), fd_base_at(codes, pos - 1) ^ 2, fp, bloom, f_b, c_b);   // a window precedes: facing backward
    return (f_f ? 1u : 0u) | (f_b ? 2u : 0u) | ((uint32_t)(c_f & 3) << 2) | ((uint32_t)(c_b & 3) << 4);
}

// every half-step in [t0, t1) is about to be scanned: evaluate the junction tests the preview left out; true if there were any
template <int MODE>
__device__ __forceinline__ bool fill_missing(WalkCtx& wc, PieceView& v, int t0, int t1) {
    if (t1 <= t0) return false;
    uint32_t qa = (uint32_t)(t0 >> 1), qb = (uint32_t)((t1 - 1) >> 1) + 1;
    if (qb > v.nwin) qb = v.nwin;
    if (pv_popc(v, v.nd0, v.nd1, wc.pl.need, qa, qb) == qb - qa) return false;   // the usual case: the preview covered the stretch
    if (MODE == WALK_PROBE || MODE == WALK_COMMIT) { wc.fail = 3; return false; }   // left to the cluster's ordered walk
    bool any = false;
    while (qa < qb) {
        const uint32_t c = qa >> 6;
        const uint32_t n = min(qb - qa, 64u - (qa & 63));
        uint64_t range = (n == 64 ? ~0ULL : ((1ULL << n) - 1)) << (qa & 63);
        uint64_t missing = range & ~pv_word(v, v.nd0, v.nd1, wc.pl.need, c);
        while (missing) {
            const uint32_t b = (uint32_t)__builtin_ctzll(missing);
            missing &= missing - 1;
            const uint32_t q = c * 64 + b;
            const uint32_t r = walk_fill_flags(wc.pl.codes, wc.bloom, wc.fp, v.p0 + q, q + 1 < v.nwin, q > 0);
            wc.n_filled++;
            any = true;
            // A junction test that comes out TRUE here is a place where this piece may create a junction, and the window's dependency
            // clusters were built without knowing that (only previewed flags are registered as candidates).  If the k-mer is a registered
            // one all the same (another position's flag, a junction of the map: the lk bit), every piece that holds it is in this cluster and
            // looks it up live: nothing is lost.  Otherwise the walk goes on -- exact as long as NO OTHER position of the window holds that
            // k-mer: no other piece reads or writes its record during this window, and later windows learn of it through the created-keys
            // lists like of any other new junction -- and notes the position; k_delta_collect, which passes over the window's positions right
            // after the walk anyway, looks for the k-mer's hash elsewhere in the window and only then voids the scan (error bit 4: the
            // library scans its journal again with every test evaluated).  Config 4's 2*10^10 positions meet this case about once per run;
            // a window there covers the genome 0.1x, so a second occurrence in the same window is the exception.
            // (The key-ordered walk voids the scan on any late test that comes out true: its cursor may already have passed the position --
            // taken and given back the k-mer's turn -- when the stretch before the stop it had chosen turns out to hold an earlier one.)
            if ((r & 3) && MODE == WALK_OVW) {
                wc.fail = 4;            // the optimistic walk only notes it: a path that does not settle voids nothing (k_ovw_commit looks at the settled one)
            } else if ((r & 3) && (MODE == WALK_KO || !((pv_word(v, v.lk0, v.lk1, wc.pl.lk, c) >> b) & 1ULL))) {
                bool noted = false;
                if (MODE == WALK_SEQ) {
                    const unsigned long long at = atomicAdd(&wc.cnt->late_n[0], 1ULL);
                    if (at < FGPU_LATE_CAP) {
                        wc.cnt->late[2 * at] = v.p0 + q;
                        wc.cnt->late[2 * at + 1] = wc.win_seq;
                        noted = true;
                    }
                }
                if (!noted) atomicOr(&wc.cnt->error_flags, 4ULL);     // (the key-ordered walk stops only at registered k-mers)
            }
            if (q >= 128) {   // these words are read from memory (pv_word): publish there
                const unsigned long long gm = 1ULL << ((v.p0 + q) & 63);
                const uint64_t gw = (v.p0 + q) >> 6;
                if (r & 1) atomicOr((unsigned long long*)&wc.pl.ff[gw], gm);
                if (r & 2) atomicOr((unsigned long long*)&wc.pl.fb[gw], gm);
                if (r & 4) atomicOr((unsigned long long*)&wc.pl.cf0[gw], gm);
                if (r & 8) atomicOr((unsigned long long*)&wc.pl.cf1[gw], gm);
                if (r & 16) atomicOr((unsigned long long*)&wc.pl.cb0[gw], gm);
                if (r & 32) atomicOr((unsigned long long*)&wc.pl.cb1[gw], gm);
                atomicOr((unsigned long long*)&wc.pl.need[gw], gm);
            } else {          // the register copies this walk works from
                const uint64_t bm = 1ULL << b;
                if (c == 0) {
                    v.nd0 |= bm;
                    if (r & 1) v.fF0 |= bm;
                    if (r & 2) v.fB0 |= bm;
                    if (r & 4) v.c0F0 |= bm;
                    if (r & 8) v.c1F0 |= bm;
                    if (r & 16) v.c0B0 |= bm;
                    if (r & 32) v.c1B0 |= bm;
                } else {
                    v.nd1 |= bm;
                    if (r & 1) v.fF1 |= bm;
                    if (r & 2) v.fB1 |= bm;
                    if (r & 4) v.c0F1 |= bm;
                    if (r & 8) v.c1F1 |= bm;
                    if (r & 16) v.c0B1 |= bm;
                    if (r & 32) v.c1B1 |= bm;
                }
            }
        }
        qa += n;
    }
    return any;
}

// A junction record held in two registers: dist[0..4] bytes 0-4, cov[0..3] bytes 5-8, linked mask byte 9.
struct RecRegs {
    uint64_t lo, hi;
    uint64_t* addr;
};
__device__ __forceinline__ uint32_t rr_get(const RecRegs& r, int i) { return (uint32_t)((i < 8 ? r.lo >> (8 * i) : r.hi >> (8 * (i - 8))) & 0xFF); }
__device__ __forceinline__ void rr_set(RecRegs& r, int i, uint32_t val) {
    if (i < 8) r.lo = (r.lo & ~(0xFFULL << (8 * i))) | ((uint64_t)val << (8 * i));
    else r.hi = (r.hi & ~(0xFFULL << (8 * (i - 8)))) | ((uint64_t)val << (8 * (i - 8)));
}
__device__ __forceinline__ void rr_update(RecRegs& r, int idx, int length) {   // Junction::update, Junction.cpp:69-71 (narrowing to uchar)
    uint32_t l = (uint32_t)length & 0xFF;
    if (rr_get(r, idx) < l) rr_set(r, idx, l);
}
__device__ __forceinline__ void rr_add_cov(RecRegs& r, int nuc) {               // Junction::addCoverage, Junction.cpp:59-67
    uint32_t c = (rr_get(r, 5 + nuc) + 1) & 0xFF;
    rr_set(r, 5 + nuc, c == 0 ? 255 : c);
}
__device__ __forceinline__ void rr_link(RecRegs& r, int idx) { rr_set(r, 9, rr_get(r, 9) | (1u << idx)); }
__device__ __forceinline__ void rr_store(const RecRegs& r) { r.addr[0] = r.lo; r.addr[1] = r.hi; }
// the key-ordered walk hands records from thread to thread inside one launch: 8-byte agent-scope accesses on both sides
__device__ __forceinline__ void st_agent(uint64_t* p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <int MODE> __device__ __forceinline__ void rr_store_m(const RecRegs& r) {
    if (MODE == WALK_KO) { st_agent(&r.addr[0], r.lo); st_agent(&r.addr[1], r.hi); }
    else rr_store(r);
}

// find or create the junction keyed by the oriented k-mer `key`; the record comes back in registers
template <int MODE>
__device__ __forceinline__ bool junction_get(WalkCtx& wc, uint64_t key, uint64_t stamp, uint64_t pos, RecRegs& out, uint64_t known_slot = ~0ULL) {
    uint64_t rc = fd_revcomp(key, wc.fp.k);
    uint64_t canon = key < rc ? key : rc;
    int orient = key == canon ? 0 : 1;
    uint64_t slot;
    uint32_t present;
    if (MODE == WALK_KO && known_slot != ~0ULL) {   // the key-ordered walk has just found this junction in the map: no second probe
        KO_T0();
        out.addr = (uint64_t*)(wc.jt.recs + (known_slot * 2 + orient) * 16);
        out.lo = ld_agent(&out.addr[0]);
        out.hi = ld_agent(&out.addr[1]);
        wc.created_now = false;
        if (out.lo == 0xFFFFFFFFFFFFFFF1ULL) return false;
        KO_T1(wc.cnt, 3);
        return true;
    }
    // the key word and the record of the home slot are requested together: at the table's low load factor the key is
    // almost always in its home slot, so an event costs one memory round trip instead of two
    const uint64_t home = fd_mix(canon) & wc.jt.mask;
    const uint64_t* spec = (const uint64_t*)(wc.jt.recs + (home * 2 + orient) * 16);
    const uint64_t w_first = ld_agent(&wc.jt.keys[home]);
    const uint64_t spec_lo = MODE == WALK_KO ? ld_agent(&spec[0]) : spec[0], spec_hi = MODE == WALK_KO ? ld_agent(&spec[1]) : spec[1];
    if (!jt_find_or_claim(wc.jt, canon, home, w_first, slot, present, wc.cnt)) {
        atomicOr(&wc.cnt->error_flags, 1ULL);
        return false;
    }
    out.addr = (uint64_t*)(wc.jt.recs + (slot * 2 + orient) * 16);
    wc.created_now = false;
    if (!((present >> orient) & 1u)) {   // JunctionMap::createJunction, JunctionMap.cpp:567-570
        wc.created_now = true;
        atomicOr(&wc.pl.cr[pos >> 6], 1ULL << (pos & 63));   // later windows (and the next batch) register this key: their snapshot cannot know it
        out.lo = out.hi = 0;
        if (MODE == WALK_KO) st_agent(&wc.jt.stamps[slot * 2 + orient], stamp);
        else wc.jt.stamps[slot * 2 + orient] = stamp;
        atomicOr((unsigned long long*)&wc.jt.keys[slot], 1ULL << (62 + orient));
        // presence filter in front of the table (phase A of later windows tests it before probing)
        const uint64_t hb = jt_filter_bit(wc.jt, canon);
        atomicOr(&wc.jt.filter[hb >> 5], 1u << (hb & 31));
        wc.n_created++;
    } else if (slot == home) {
        out.lo = spec_lo;
        out.hi = spec_hi;
    } else {
        out.lo = MODE == WALK_KO ? ld_agent(&out.addr[0]) : out.addr[0];
        out.hi = MODE == WALK_KO ? ld_agent(&out.addr[1]) : out.addr[1];
    }
    return true;
}


// the record of an EXISTING junction, read-only (WALK_PROBE / WALK_COMMIT); false = the key is not in the map (a walk would create it)
__device__ __forceinline__ bool junction_find(const WalkCtx& wc, uint64_t key, RecRegs& out) {
    const uint64_t rc = fd_revcomp(key, wc.fp.k);
    const uint64_t canon = key < rc ? key : rc;
    const int orient = key == canon ? 0 : 1;
    uint64_t slot;
    uint32_t present;
    if (!jt_find_live(wc.jt, canon, slot, present) || !((present >> orient) & 1u)) return false;
    out.addr = (uint64_t*)(wc.jt.recs + (slot * 2 + orient) * 16);
    out.lo = out.addr[0];
    out.hi = out.addr[1];
    return true;
}
__device__ __forceinline__ bool rr_raises(const RecRegs& r, int idx, int length) { return rr_get(r, idx) < ((uint32_t)length & 0xFF); }
// Junction::addCoverage / the link flags on a record other threads update at the same time (WALK_COMMIT): saturating +1 on one byte,
// OR of one bit; `seen` is this thread's earlier plain read of the record and only saves atomics that cannot change anything
__device__ __forceinline__ void rec_add_cov_atomic(const RecRegs& seen, int nuc) {
    if (rr_get(seen, 5 + nuc) == 255) return;
    unsigned long long* w = (unsigned long long*)(seen.addr + (nuc < 3 ? 0 : 1));
    const int sh = nuc < 3 ? 8 * (5 + nuc) : 0;
    unsigned long long old = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (((old >> sh) & 0xFF) != 255) {
        const unsigned long long was = atomicCAS(w, old, old + (1ULL << sh));
        if (was == old) break;
        old = was;
    }
}
__device__ __forceinline__ void rec_link_atomic(const RecRegs& seen, int idx) {
    if ((rr_get(seen, 9) >> idx) & 1u) return;
    atomicOr((unsigned long long*)(seen.addr + 1), 1ULL << (8 + idx));
}


// -DFGPU_KO_TRACE: one piece in 16 also leaves a time stamp at every step of every position it deals with (kt.trace + 2^23 words on:
// 1024 words per stamped piece: [0] piece << 16 | stamps, then {code << 56 | q << 40 | ticks since the piece's start}; scripts/ko_trace.py)
#ifdef FGPU_KO_TRACE
#define KO_STAMP(ko, code, q)                                                                                                         \
    do {                                                                                                                              \
        if ((ko).stamp_base && (ko).stamp_n < 1020)                                                                                    \
            (ko).stamp_base[1 + (ko).stamp_n] = ((unsigned long long)(code) << 56) | ((unsigned long long)(q) << 40) |                 \
                                                ((wall_clock64() - (ko).stamp_t0) & 0xFFFFFFFFFFULL);                                  \
        if ((ko).stamp_base) (ko).stamp_n++;                                                                                          \
    } while (0)
#else
#define KO_STAMP(ko, code, q) do {} while (0)
#endif
// ---- turn taking of the key-ordered walk (see KoTables) ---------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ko_ordinal(const KoState& ko, uint32_t q) {   // lk positions of the piece below q
    (void)ko;
    uint32_t n = 0;
    for (uint32_t c = 0; c < (q >> 6); c++) n += (uint32_t)__popcll(ko_lk(c));
    return n + (uint32_t)__popcll(ko_lk(q >> 6) & ((1ULL << (q & 63)) - 1));
}
__device__ __forceinline__ bool ko_is_lk(const KoState& ko, uint32_t q) { (void)ko; return ((ko_lk(q >> 6) >> (q & 63)) & 1ULL) != 0; }
__device__ __noinline__ unsigned long long ko_wait(const uint32_t* turn, uint32_t r, DevCounters* cnt) {
    unsigned spins = 0;
    unsigned long long t0 = 0;
#ifdef FGPU_KO_TRACE
    if (ld_agent(turn) == r) return 0;            // (only what is spent AFTER a first look that found the turn elsewhere counts as waiting)
    const unsigned long long t_in = wall_clock64();
#endif
#ifdef FGPU_KO_TIMING
    const unsigned long long tw = wall_clock64();
    if (ld_agent(turn) != r) {
        while (ld_agent(turn) != r) __builtin_amdgcn_s_sleep(1);
        atomicAdd(&cnt->par_probe[1], wall_clock64() - tw);
        atomicAdd(&cnt->par_probe[3], 1ULL);
    }
    return 0;
#endif
    while (ld_agent(turn) != r) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 4095u) == 0) {   // a turn that never comes is a bug, not a state to wait in: say so once and let every piece run out
            if (ld_agent((const uint64_t*)&cnt->error_flags) & 8ULL) break;
            const unsigned long long now = wall_clock64();          // constant 100 MHz
            if (!t0) t0 = now;
            else if (now - t0 > KO_WAIT_LIMIT_TICKS) { atomicOr(&cnt->error_flags, 8ULL); break; }
        }
    }
    // no cache invalidate: everything one piece hands to the next (turn counters, key words, records) is read with agent-scope loads
#ifdef FGPU_KO_TRACE
    return wall_clock64() - t_in;
#else
    return 0;
#endif
}
__device__ __forceinline__ DevCounters* ko_cnt(const KoState& ko) { return ko.cnt; }
// the k-mer's turn goes to its next occurrence; whatever this piece stored is visible before the counter moves
__device__ __forceinline__ void ko_give(const KoState& ko, KoHold& h) {
    // records and stamps are stored with agent-scope (write-through, sc1) stores and key words / planes change by device-scope atomics, so the
    // release is: wait until every one of them has been acknowledged (gfx950 counts stores and atomics in vmcnt), then move the counter.  A
    // workgroup-scope fence alone emits NO wait on this target (the turn store followed the record stores with nothing in between: ADVICE r2);
    // an agent-scope release fence would add a write-back of the whole L2 (buffer_wbl2), which nothing here needs: no plain store is shared.
    {
        KO_T0();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");       // compiler barrier: no store may sink below the counter's
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __hip_atomic_store(&ko.kt.hk_turn[h.e], h.r + h.n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        KO_T1(ko_cnt(ko), 1);
    }
    h.n = 0;
}
// the cursor leaves its position without having visited it
__device__ __forceinline__ void ko_leave_cursor(KoState& ko, bool have_last) {
    if (!ko.cur.n) return;
    if (ko.cur_q == ko.mid && !have_last && !ko.fake.n) { ko.fake = ko.cur; ko.cur.n = 0; }   // add_fake_junction may come back to it
    else ko_give(ko, ko.cur);
}
// account for the lk position q: take its turn; keep it (cursor) or pass it on
__device__ __forceinline__ void ko_account(WalkCtx& wc, uint32_t q, bool as_cursor, bool have_last) {
    KoState& ko = wc.ko;
    ko_leave_cursor(ko, have_last);
    KO_T0();
    KO_STAMP(ko, 1, q);
    const uint32_t node = ko.base + ko_ordinal(ko, q);
    const uint32_t e = ko.kt.occ_entry[node], r = ko.kt.occ_rank[node];
    if (as_cursor) ko.cur_q = q;
    if (e == 0xFFFFFFF0u) return;                   // (keeps the loads in front of the clock)
    KO_T1(wc.cnt, 0);
    if (ko.last.n && e == ko.last.e) { ko.last.n++; if (as_cursor) ko.cur_in = 1; return; }     // this piece holds the k-mer already
    if (ko.fake.n && e == ko.fake.e) { ko.fake.n++; if (as_cursor) ko.cur_in = 2; return; }
    ko.wait_acc += ko_wait(&ko.kt.hk_turn[e], r, wc.cnt);
    KO_STAMP(ko, 2, q);
    KoHold h;
    h.e = e; h.r = r; h.n = 1;
    if (as_cursor) { ko.cur = h; ko.cur_in = 0; }
    else if (q == ko.mid && !have_last && !ko.fake.n) ko.fake = h;
    else ko_give(ko, h);
}
// every lk position below q_to has been passed
__device__ __forceinline__ void ko_pass(WalkCtx& wc, uint32_t q_to, bool have_last) {
    KoState& ko = wc.ko;
    while (ko.done < q_to) {
        if (ko.done >= 64 * KO_CHUNKS) { ko.done = q_to; break; }
        const uint64_t w = ko_lk(ko.done >> 6) >> (ko.done & 63);
        if (!w) {                                   // no lk position in the rest of this word
            const uint32_t next = ((ko.done >> 6) + 1) * 64;
            ko.done = next < q_to ? next : q_to;
            continue;
        }
        const uint32_t q = ko.done + (uint32_t)__builtin_ctzll(w);
        if (q >= q_to) { ko.done = q_to; break; }
        ko_account(wc, q, false, have_last);
        ko.done = q + 1;
    }
}
// the walk is about to look at (or stop at) position q
__device__ __forceinline__ void ko_cursor(WalkCtx& wc, uint32_t q, bool have_last) {
    KoState& ko = wc.ko;
    if (q < ko.done) return;                       // second half-step of the same position: its k-mer is still held
    ko_pass(wc, q, have_last);
    if (ko_is_lk(ko, q)) ko_account(wc, q, true, have_last);
    else { ko_leave_cursor(ko, have_last); atomicOr(&wc.cnt->error_flags, 16ULL); }   // a stop at a k-mer nobody registered: cannot happen
    ko.done = q + 1;
}
// the junction under the cursor has been visited and becomes the piece's last junction; the previous one's record has been stored
__device__ __forceinline__ void ko_visited(WalkCtx& wc) {
    KoState& ko = wc.ko;
    if (ko.cur_in == 0) {
        if (ko.last.n) ko_give(ko, ko.last);
        ko.last = ko.cur;
        ko.cur.n = 0;
    } else if (ko.cur_in == 2) {
        if (ko.last.n) ko_give(ko, ko.last);
        ko.last = ko.fake;
        ko.fake.n = 0;
    }
    ko.cur_in = 1;
    if (ko.fake.n) ko_give(ko, ko.fake);           // a junction was found: there will be no fake one
}
// end of the piece (also after an error): everything is passed, every turn given back
__device__ __forceinline__ void ko_finish(WalkCtx& wc, uint32_t nwin) {
    KoState& ko = wc.ko;
    ko_pass(wc, nwin < 64 * KO_CHUNKS ? nwin : 64 * KO_CHUNKS, true);
    if (ko.cur.n) ko_give(ko, ko.cur);
    if (ko.last.n) ko_give(ko, ko.last);
    if (ko.fake.n) ko_give(ko, ko.fake);
}

// scan_forward (ReadScanner.cpp:112-206) for the piece {p0, nwin}
template <int MODE>
__device__ __forceinline__ void walk_piece(WalkCtx& wc, uint64_t p0, uint32_t nwin, uint64_t piece_seq) {
    const int k = wc.fp.k, j = wc.fp.j;
    const int tmax = 2 * (int)nwin - 2 - 2 * j;     // last half-step with distToEnd > 2j
    const int spacer = 2 * wc.fp.max_spacer - 1;
    PieceView v;
    pv_load(v, wc.pl, p0, nwin);
    if (MODE != WALK_KO) {
        created_bits(wc, v, 0, v.xF0, v.xB0);       // what the snapshot planes cannot know: the live table at the candidate positions
        if (nwin > 64) created_bits(wc, v, 1, v.xF1, v.xB1);
    }
    int t = 2 * j + 1;
    int last_pos = 0;                               // lastJuncPos
    bool have_last = false;
    RecRegs last;
    last.lo = last.hi = 0;
    last.addr = nullptr;
    int last_t = 0, last_ext_fwd = 0;

    while (t <= tmax) {
        // ---- find_next_junction (ReadScanner.cpp:61-86): first t' >= t that is in the map, hits the spacer rule, or is flagged
        int t_sp = last_pos + spacer;
        if (t_sp < t) t_sp = t;
        int tn;
        uint32_t q = 0;
        bool fwd = false, in_map = false, by_spacer = false;
        uint64_t ko_slot = ~0ULL;
        for (;;) {   // repeated when junction tests had to be evaluated on the spot (fill_missing)
            int t_ev = 0x7fffffff;
            bool ev_in_map = false;
            {
                const uint32_t q0 = (uint32_t)(t >> 1);
                const int t_stop = tmax < t_sp ? tmax : t_sp;
                for (uint32_t c = q0 >> 6; c * 64 < nwin && 2 * (int)(c * 64) <= t_stop; c++) {
                    uint64_t mF, mB;
                    in_map_words<MODE>(wc, v, c, mF, mB);
                    uint64_t eF = mF | pv_word(v, v.fF0, v.fF1, wc.pl.ff, c);
                    uint64_t eB = mB | pv_word(v, v.fB0, v.fB1, wc.pl.fb, c);
                    if (c == (q0 >> 6)) {               // nothing before q0; at q0 the backward half-step is behind us if t is odd
                        const uint64_t from = ~0ULL << (q0 & 63);
                        eF &= from;
                        eB &= from;
                        if (t & 1) eB &= ~(1ULL << (q0 & 63));
                    }
                    int tb = eB ? 2 * (int)(c * 64 + __builtin_ctzll(eB)) : 0x7fffffff;
                    int tf = eF ? 2 * (int)(c * 64 + __builtin_ctzll(eF)) + 1 : 0x7fffffff;
                    int te = tb < tf ? tb : tf;
                    if (te != 0x7fffffff) {
                        t_ev = te;
                        ev_in_map = ((te & 1) ? mF : mB) >> ((te >> 1) & 63) & 1ULL;
                        break;
                    }
                }
            }
            tn = t_ev < t_sp ? t_ev : t_sp;
            if (tn > tmax) {   // runs off the end of the piece: everything up to tmax is scanned
                if (fill_missing<MODE>(wc, v, t, tmax + 1)) continue;
                break;
            }
            q = (uint32_t)(tn >> 1);
            fwd = tn & 1;
            // why did we stop here?  (order of the tests in find_next_junction)
            if (tn == t_ev) {
                in_map = ev_in_map;
            } else {           // stopped by the spacer rule before any event: the position itself may still be in the map
                uint64_t mF, mB;
                in_map_words<MODE>(wc, v, q >> 6, mF, mB);
                in_map = ((fwd ? mF : mB) >> (q & 63)) & 1ULL;
            }
            if (MODE == WALK_KO) {   // the k-mer's turn first; then the map is asked, and says what a piece walked in file order would see
                KO_STAMP(wc.ko, 4, q);
                ko_cursor(wc, q, have_last);
                const bool potential = in_map;
                const uint64_t kmq = pv_kmer(v, wc.pl.codes, p0 + q, k);
                const uint64_t rcq = fd_revcomp(kmq, k);
                const uint64_t canon = kmq < rcq ? kmq : rcq;
                uint64_t slot = 0;      // (was compared uninitialised when the k-mer is not in the table: undefined behaviour that one
                uint32_t present = 0;   // build of round 3 did not survive -- "a k-mer's turn never came" at full size)
                {
                    KO_T0();
                    in_map = jt_find_live(wc.jt, canon, slot, present) && ((present >> ((fwd ? kmq : rcq) == canon ? 0 : 1)) & 1u);
                    if (slot == 0xFFFFFFFFFFFFFFF0ULL) return;
                    KO_T1(wc.cnt, 2);
                }
                KO_STAMP(wc.ko, 3, q);
                ko_slot = in_map ? slot : ~0ULL;
                if (potential && !in_map) {                 // registered, but not in the map (yet): not an event; look again from here
                    ko_absent(fwd, q >> 6) |= 1ULL << (q & 63);
                    continue;
                }
            }
            by_spacer = !in_map && (tn - last_pos >= spacer);
            if (fill_missing<MODE>(wc, v, t, (in_map || by_spacer) ? tn : tn + 1)) continue;
            break;
        }
        if ((MODE == WALK_PROBE || MODE == WALK_COMMIT) && wc.fail) return;
        if (tn > tmax) {   // ran off the end of the piece
            wc.nb_processed += (unsigned long long)(tmax - t + 1);
            wc.nb_jcheck += jcheck_sum(wc, v, t, tmax + 1);
            break;
        }
        wc.nb_processed += (unsigned long long)(tn - t);
        wc.nb_jcheck += jcheck_sum(wc, v, t, (in_map || by_spacer) ? tn : tn + 1);

        // ---- junction at (q, fwd)  (ReadScanner.cpp:133-192)
        uint64_t km = pv_kmer(v, wc.pl.codes, p0 + q, k);
        uint64_t key = fwd ? km : fd_revcomp(km, k);
        int real = fwd ? pv_base(v, wc.pl.codes, p0 + q + k) : (pv_base(v, wc.pl.codes, p0 + q - 1) ^ 2);
        RecRegs cur;
        const int ext_fwd = fwd ? real : 4;          // getExtensionIndex(FORWARD)
        const int ext_bwd = fwd ? 4 : real;          // getExtensionIndex(BACKWARD)
        if (MODE == WALK_SEQ || MODE == WALK_KO) {
            if (!junction_get<MODE>(wc, key, (piece_seq << STAMP_SHIFT) | (uint64_t)tn, p0 + q, cur, ko_slot)) return;
            if (wc.pl.sF) atomicOr(&(fwd ? wc.pl.sF : wc.pl.sB)[(p0 + q) >> 6], 1ULL << ((p0 + q) & 63));   // result.push_back, :140
            if (MODE == WALK_SEQ && wc.created_now) {   // the new key may recur further along this piece (tandem repeats)
                created_bits(wc, v, 0, v.xF0, v.xB0);
                if (nwin > 64) created_bits(wc, v, 1, v.xF1, v.xB1);
            }
            const bool same = have_last && cur.addr == last.addr;   // the same junction twice in a row: one register copy
            if (same) cur = last;
            last_pos = tn;
            rr_add_cov(cur, real);
            if (have_last) {                             // directLinkJunctions, JunctionMap.cpp:551-561
                const int d = tn - last_t;
                if (same) {
                    rr_update(cur, last_ext_fwd, d);
                    rr_link(cur, last_ext_fwd);
                } else {
                    rr_update(last, last_ext_fwd, d);
                    rr_link(last, last_ext_fwd);
                    rr_store_m<MODE>(last);
                }
                rr_update(cur, ext_bwd, d);
                rr_link(cur, ext_bwd);
            } else {
                have_last = true;
                rr_update(cur, ext_bwd, tn - 2 * j);
            }
        } else {
            // Out of order: the junction must exist already, and no distance may be raised (a later piece's skips read them) -- then the
            // only things this visit changes are a coverage count and two link flags, which no walk reads.
            if (!junction_find(wc, key, cur)) { wc.fail = 1; return; }
            if (MODE == WALK_COMMIT) {
                if (wc.pl.sF) atomicOr(&(fwd ? wc.pl.sF : wc.pl.sB)[(p0 + q) >> 6], 1ULL << ((p0 + q) & 63));
                rec_add_cov_atomic(cur, real);
            }
            last_pos = tn;
            if (have_last) {
                const int d = tn - last_t;
                if (MODE == WALK_PROBE) {
                    if (rr_raises(last, last_ext_fwd, d) || rr_raises(cur, ext_bwd, d)) { wc.fail = 2; return; }
                } else {
                    rec_link_atomic(last, last_ext_fwd);
                    rec_link_atomic(cur, ext_bwd);
                }
            } else {
                have_last = true;
                if (MODE == WALK_PROBE && rr_raises(cur, ext_bwd, tn - 2 * j)) { wc.fail = 2; return; }
            }
        }
        last = cur;
        last_t = tn;
        last_ext_fwd = ext_fwd;
        int d = (int)rr_get(cur, ext_fwd);
        if (d < 1) d = 1;
        t = tn + d;
        wc.nb_processed += 1;
        wc.nb_skipped += (unsigned long long)(d - 1);
        if (MODE == WALK_KO) {   // the previous junction's record is stored: its k-mer goes on; so do the positions the skip jumps over
            KO_STAMP(wc.ko, 5, q);
            ko_visited(wc);
            const uint32_t q_next = (uint32_t)(t >> 1);
            ko_pass(wc, q_next < nwin ? q_next : nwin, true);
            KO_STAMP(wc.ko, 6, q);
        }
    }

    if (!have_last) {   // add_fake_junction (ReadScanner.cpp:92-104)
        wc.nb_no_juncs++;
        const int len = (int)nwin + k - 1;
        const int m = len / 2 - k / 2;
        uint64_t key = pv_kmer(v, wc.pl.codes, p0 + m, k);
        int real = pv_base(v, wc.pl.codes, p0 + m + k);
        RecRegs rec;
        const int tm = 2 * m + 1;
        if (MODE == WALK_SEQ || MODE == WALK_KO) {
            if (MODE == WALK_KO) {   // the middle k-mer's turn was kept when the cursor passed it (ko.fake), or is taken now
                ko_leave_cursor(wc.ko, false);
                ko_pass(wc, (uint32_t)m + 1, false);
            }
            if (!junction_get<MODE>(wc, key, (piece_seq << STAMP_SHIFT) | STAMP_FAKE, p0 + (uint64_t)m, rec)) return;
            rr_add_cov(rec, real);
            rr_update(rec, 4, tm - 2 * j);
            rr_update(rec, real, (2 * (int)nwin - 1 - tm) - 2 * j);
            rr_store_m<MODE>(rec);
        } else {
            if (!junction_find(wc, key, rec)) { wc.fail = 1; return; }
            if (MODE == WALK_PROBE) {
                if (rr_raises(rec, 4, tm - 2 * j) || rr_raises(rec, real, (2 * (int)nwin - 1 - tm) - 2 * j)) wc.fail = 2;
            } else {
                rec_add_cov_atomic(rec, real);
            }
        }
    } else {            // ReadScanner.cpp:202-206
        if (MODE == WALK_SEQ || MODE == WALK_KO) {
            rr_update(last, last_ext_fwd, (2 * (int)nwin - 1 - last_t) - 2 * j);
            rr_store_m<MODE>(last);
        } else if (MODE == WALK_PROBE && rr_raises(last, last_ext_fwd, (2 * (int)nwin - 1 - last_t) - 2 * j)) {
            wc.fail = 2;
        }
    }
}

constexpr uint32_t LOCAL_MEMBERS = 16;
__global__ void __launch_bounds__(64) k_walk(Planes pl, FdParams fp, JTable jt, const uint32_t* __restrict__ root,
                                             const uint32_t* __restrict__ count, const uint32_t* __restrict__ head,
                                             const uint32_t* __restrict__ next, uint32_t* pool, const WinDesc* __restrict__ wdp,
                                             uint64_t piece_seq_base, const uint32_t* __restrict__ bloom, DevCounters* cnt, int dbg,
                                             const uint32_t* __restrict__ par_fail, uint32_t heavy, const uint32_t* __restrict__ ko_bad,
                                             const uint32_t* __restrict__ ko_state, uint32_t ko_heavy, uint32_t win_seq, uint32_t ko_heavy_w) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    WalkCtx wc;
    wc.pl = pl; wc.fp = fp; wc.jt = jt; wc.cnt = cnt; wc.bloom = bloom; wc.win_seq = win_seq;
    wc.nb_processed = wc.nb_skipped = wc.nb_jcheck = wc.nb_no_juncs = wc.n_created = wc.n_filled = 0;
    wc.created_now = false; wc.fail = 0; wc.dbg = dbg; 
    const WinDesc wd = *wdp;
    const uint32_t n = wd.n, first_piece = wd.first_piece;
    // the large-cluster tables did not hold this window (its clusters are walked here, in order): the host shrinks the windows that follow
    if (ko_heavy && blockIdx.x == 0 && threadIdx.x == 0 && (ko_state[1] & 1u)) atomicAdd(&cnt->ko_overflows, 1ULL);
    unsigned long long n_follow = 0, biggest = 0;
    // everything the thread needs to start is requested at once (no dependent loads before the plane burst)
    const uint32_t ii = i < n ? i : 0;
    const uint32_t my_root = root[ii], my_count = count[ii], my_head = head[ii];
    const uint2 my_piece = pl.pieces[first_piece + ii];

    // a large cluster whose pieces k_walk_par found to be order-free is walked there, one thread per piece
    const bool walked_out_of_order = (heavy && my_count + 1 >= heavy && !par_fail[ii]) ||
                                     (ko_heavy && ko_cluster(my_count, par_fail[ii], ko_heavy, ko_heavy_w) && !ko_bad[ii] && !(ko_state[1] & 1u));   // k_walk_ko has them (par_fail = the weights then)
    if (i < n && my_root == i && !walked_out_of_order) {
        const uint32_t nm = (dbg & 2) ? 0 : my_count;
        uint32_t local_mem[LOCAL_MEMBERS];
        uint32_t* mem = local_mem;
        if (nm) {   // followers: off the linked list, into ascending piece order (the leader is the smallest index of the cluster)
            n_follow = nm;
            biggest = nm + 1;
            if (nm > LOCAL_MEMBERS) mem = pool + atomicAdd(&cnt->pad2, (unsigned long long)nm);   // rare: giant cluster, list in global memory
            uint32_t j = 0;
            for (uint32_t m = my_head; m != U_INF && j < nm; m = next[m]) mem[j++] = m;
            for (uint32_t gap = nm / 2; gap > 0; gap /= 2)          // shell sort: fine for 2 members and for 10^5
                for (uint32_t a = gap; a < nm; a++) {
                    uint32_t v = mem[a];
                    uint32_t b = a;
                    while (b >= gap && mem[b - gap] > v) { mem[b] = mem[b - gap]; b -= gap; }
                    mem[b] = v;
                }
        }
        for (uint32_t a = 0; a <= nm; a++) {
            const uint32_t m = a == 0 ? i : mem[a - 1];
            const uint2 pc = a == 0 ? my_piece : pl.pieces[first_piece + m];
            walk_piece<WALK_SEQ>(wc, pc.x, pc.y, piece_seq_base + first_piece + m);
        }
    }
    // wave-level reduction of the counters
    unsigned long long v[7] = {wc.nb_processed, wc.nb_skipped, wc.nb_jcheck, wc.nb_no_juncs, wc.n_created, n_follow, wc.n_filled};
    for (int c = 0; c < 7; c++)
        for (int o = 32; o > 0; o >>= 1) v[c] += __shfl_down(v[c], o, 64);
    for (int o = 32; o > 0; o >>= 1) { unsigned long long t = __shfl_down(biggest, o, 64); biggest = t > biggest ? t : biggest; }
    if (fd_lane() == 0) {
        if (v[5]) atomicAdd(&cnt->followers, v[5]);
        if (biggest) atomicMax(&cnt->max_cluster, biggest);
        if (v[0]) atomicAdd(&cnt->nb_processed, v[0]);
        if (v[1]) atomicAdd(&cnt->nb_skipped, v[1]);
        if (v[2]) atomicAdd(&cnt->nb_jcheck, v[2]);
        if (v[3]) atomicAdd(&cnt->nb_no_juncs, v[3]);
        if (v[4]) atomicAdd(&cnt->n_junctions, v[4]);
        if (v[6]) atomicAdd(&cnt->flags_filled, v[6]);
    }
}

// The same walk with the clusters handed out DYNAMICALLY (round 4).  k_walk gives lane i the cluster led by piece i: lanes whose piece is a
// follower idle, lanes with a small cluster idle while a neighbour walks a large one, and a wave lives as long as its longest cluster --
// 7.6 of 64 lanes active per vector instruction on config 2, 13 on config 4's thin windows, a fifth of the wave slots occupied
// (profiles/r04_walk_counters.txt).  Here the waves of a fixed grid draw clusters from the window's list of leaders (k_walk_cluster) whenever a
// lane has none, one reservation per wave and round, and every round every lane that holds a cluster walks ONE piece of it: the lanes stay
// aligned piece by piece, a finished lane is refilled at once, and the grid drains when the list does.  Clusters are independent of each
// other (that is what makes them clusters), so the order in which they are handed out changes nothing.
__global__ void __launch_bounds__(64) k_walk_dyn(Planes pl, FdParams fp, JTable jt, const uint32_t* __restrict__ root_list, uint32_t* roots_state,
                                                 const uint32_t* __restrict__ count, const uint32_t* __restrict__ head,
                                                 const uint32_t* __restrict__ next, uint32_t* pool, const WinDesc* __restrict__ wdp,
                                                 uint64_t piece_seq_base, const uint32_t* __restrict__ bloom, DevCounters* cnt, int dbg,
                                                 const uint32_t* __restrict__ par_fail, uint32_t heavy, const uint32_t* __restrict__ ko_bad,
                                                 const uint32_t* __restrict__ ko_state, uint32_t ko_heavy, uint32_t win_seq, uint32_t ko_heavy_w) {
    WalkCtx wc;
    wc.pl = pl; wc.fp = fp; wc.jt = jt; wc.cnt = cnt; wc.bloom = bloom; wc.win_seq = win_seq;
    wc.nb_processed = wc.nb_skipped = wc.nb_jcheck = wc.nb_no_juncs = wc.n_created = wc.n_filled = 0;
    wc.created_now = false; wc.fail = 0; wc.dbg = dbg;
    const WinDesc wd = *wdp;
    const uint32_t first_piece = wd.first_piece;
    if (ko_heavy && blockIdx.x == 0 && threadIdx.x == 0 && (ko_state[1] & 1u)) atomicAdd(&cnt->ko_overflows, 1ULL);
    const uint32_t n_roots = roots_state[0];
    const bool ko_usable = ko_heavy && !(ko_state[1] & 1u);
    unsigned long long n_follow = 0, biggest = 0;
    uint32_t local_mem[LOCAL_MEMBERS];
    uint32_t* mem = local_mem;
    uint32_t leader = U_INF, nm = 0, at = 0;          // the lane's cluster: its leader, followers, next piece (0 = the leader itself)
    bool drained = false;
    for (;;) {
        // lanes without a cluster draw the next leaders of the list
        const bool want = leader == U_INF && !drained;
        const unsigned long long wm = __ballot(want);
        if (wm) {
            const int first = __builtin_ctzll(wm);
            uint32_t base = 0;
            if (fd_lane() == first) base = atomicAdd(&roots_state[1], (uint32_t)__popcll(wm));
            base = (uint32_t)__shfl((int)base, first, 64);
            if (want) {
                const uint32_t mine = base + (uint32_t)__popcll(wm & ((1ULL << fd_lane()) - 1));
                if (mine >= n_roots) drained = true;
                else {
                    const uint32_t i = root_list[mine];
                    const uint32_t my_count = count[i];
                    // a large cluster that the out-of-order / key-ordered / optimistic walks take is left alone here
                    const bool elsewhere = (heavy && my_count + 1 >= heavy && !par_fail[i]) ||
                                           (ko_usable && ko_cluster(my_count, par_fail[i], ko_heavy, ko_heavy_w) && !ko_bad[i]);
                    if (!elsewhere) {
                        leader = i;
                        at = 0;
                        nm = (dbg & 2) ? 0 : my_count;
                        mem = local_mem;
                        if (nm) {   // followers: off the linked list, into ascending piece order (the leader is the smallest index of the cluster)
                            n_follow += nm;
                            if (nm + 1 > biggest) biggest = nm + 1;
                            if (nm > LOCAL_MEMBERS) mem = pool + atomicAdd(&cnt->pad2, (unsigned long long)nm);   // rare: giant cluster, list in global memory
                            uint32_t j = 0;
                            for (uint32_t m = head[i]; m != U_INF && j < nm; m = next[m]) mem[j++] = m;
                            for (uint32_t gap = nm / 2; gap > 0; gap /= 2)          // shell sort: fine for 2 members and for 10^5
                                for (uint32_t a = gap; a < nm; a++) {
                                    uint32_t v = mem[a];
                                    uint32_t b = a;
                                    while (b >= gap && mem[b - gap] > v) { mem[b] = mem[b - gap]; b -= gap; }
                                    mem[b] = v;
                                }
                        }
                    }
                }
            }
        }
        if (!__ballot(leader != U_INF)) {
            if (!__ballot(!drained)) break;               // nobody holds a cluster and the list is exhausted for every lane
            continue;                                     // (some lane drew a cluster that is walked elsewhere: it draws again)
        }
        if (leader != U_INF) {                            // one piece of the lane's cluster
            const uint32_t m = at == 0 ? leader : mem[at - 1];
            const uint2 pc = pl.pieces[first_piece + m];
            walk_piece<WALK_SEQ>(wc, pc.x, pc.y, piece_seq_base + first_piece + m);
            if (at++ == nm) leader = U_INF;
        }
    }
    unsigned long long v[7] = {wc.nb_processed, wc.nb_skipped, wc.nb_jcheck, wc.nb_no_juncs, wc.n_created, n_follow, wc.n_filled};
    for (int c = 0; c < 7; c++)
        for (int o = 32; o > 0; o >>= 1) v[c] += __shfl_down(v[c], o, 64);
    for (int o = 32; o > 0; o >>= 1) { unsigned long long t = __shfl_down(biggest, o, 64); biggest = t > biggest ? t : biggest; }
    if (fd_lane() == 0) {
        if (v[5]) atomicAdd(&cnt->followers, v[5]);
        if (biggest) atomicMax(&cnt->max_cluster, biggest);
        if (v[0]) atomicAdd(&cnt->nb_processed, v[0]);
        if (v[1]) atomicAdd(&cnt->nb_skipped, v[1]);
        if (v[2]) atomicAdd(&cnt->nb_jcheck, v[2]);
        if (v[3]) atomicAdd(&cnt->nb_no_juncs, v[3]);
        if (v[4]) atomicAdd(&cnt->n_junctions, v[4]);
        if (v[6]) atomicAdd(&cnt->flags_filled, v[6]);
    }
}

// The out-of-order walk of large clusters.  A cluster is the set of pieces of a window that share a junction k-mer; its one thread walks
// them in file order because a piece may create junctions and raise distances that later pieces' skips depend on.  Where a repeat of the
// genome is covered a thousand times the cluster holds hundreds of pieces per window, each with a junction at every position, and that one
// thread decides how long the window takes (a 6 Mb genome with twenty copies of a 400-base repeat: 29 ms per window, 1.9 s of a 2 s scan).
// But in such a cluster nearly every piece changes nothing another piece can see: its junctions exist and their distances -- maxima over
// all earlier reads -- are not raised.  What is left of scan_forward then is a coverage count and two link flags per visit
// (Junction::addCoverage saturating at 255, linked[] = true: JunctionMap.cpp:551-561), and those commute.  So, for clusters of at least
// `heavy` pieces, one thread per PIECE:
//   PROBE  (before k_walk)  walk read-only; par_fail[root] is set if any piece of the cluster would create a junction, raise a distance,
//          or needs junction tests the preview left out.  If nobody sets it, no distance and no key of the cluster changes during this
//          window, so every piece reads exactly what it would read in file order: the paths found here are the sequential paths.
//   k_walk walks the clusters that failed (and all small ones) in order, as before.
//   COMMIT (after k_walk)   the pieces of the clusters that passed walk the same path again and apply coverage and links with atomics,
//          mark their visits for scanInputRead's lists and add their counters.
// Keys are exclusive to a cluster (that is what makes it a cluster), so the three kernels never touch each other's records.
template <int MODE>
__global__ void __launch_bounds__(64) k_walk_par(Planes pl, FdParams fp, JTable jt, const uint32_t* __restrict__ root, const uint32_t* __restrict__ count,
                                                 uint32_t* par_fail, const WinDesc* __restrict__ wdp, const uint32_t* __restrict__ bloom, DevCounters* cnt,
                                                 uint32_t heavy) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    WalkCtx wc;
    wc.pl = pl; wc.fp = fp; wc.jt = jt; wc.cnt = cnt; wc.bloom = bloom; wc.win_seq = 0;
    wc.nb_processed = wc.nb_skipped = wc.nb_jcheck = wc.nb_no_juncs = wc.n_created = wc.n_filled = 0;
    wc.created_now = false; wc.fail = 0; wc.dbg = 0; 
    const WinDesc wd = *wdp;
    unsigned long long walked = 0;
    if (i < wd.n) {
        const uint32_t r = root[i];
        if (count[r] + 1 >= heavy && (MODE == WALK_PROBE || !par_fail[r])) {
            const uint2 pc = pl.pieces[wd.first_piece + i];
            walk_piece<MODE>(wc, pc.x, pc.y, 0);
            if (MODE == WALK_PROBE) {
                if (wc.fail) par_fail[r] = 1u;
                atomicAdd(&cnt->par_probe[wc.fail], 1ULL);
            } else {
                walked = 1;
            }
        }
    }
    if (MODE == WALK_COMMIT) {
        unsigned long long v[5] = {wc.nb_processed, wc.nb_skipped, wc.nb_jcheck, wc.nb_no_juncs, walked};
        for (int c = 0; c < 5; c++)
            for (int o = 32; o > 0; o >>= 1) v[c] += __shfl_down(v[c], o, 64);
        if (fd_lane() == 0) {
            if (v[0]) atomicAdd(&cnt->nb_processed, v[0]);
            if (v[1]) atomicAdd(&cnt->nb_skipped, v[1]);
            if (v[2]) atomicAdd(&cnt->nb_jcheck, v[2]);
            if (v[3]) atomicAdd(&cnt->nb_no_juncs, v[3]);
            if (v[4]) atomicAdd(&cnt->walk_parallel, v[4]);
        }
    }
}

// (tables of the optimistic walk, k_ovw_round further down: k_ko_prepare lists its pieces, k_walk_ko looks at its outcome)
constexpr int OVW_MAX_ROUNDS = 24;
constexpr uint32_t OVW_CREATION = 5;       // event index of "the record is created" (0..4: the distances)
constexpr uint64_t OVW_EPOCH_MASK = 0xFFULL << 56;
enum : uint32_t { OVW_F_CREATED = 1, OVW_F_REAL = 6 /* bits 1-2 */, OVW_F_LINK_B = 8, OVW_F_LINK_F = 16, OVW_F_FAKE = 32, OVW_F_DF = 64, OVW_F_PRESENT = 128 };

struct OvwEv {   // the events of one round: open addressing, one entry per (key, non-dominated event); entries of another epoch are free slots
    unsigned long long* key;   // epoch << 56 | record << 3 | index
    unsigned long long* val;   // epoch << 56 | time << 8 | value
    uint32_t* bits;            // presence filter in front (by hash of record and index)
    uint64_t mask;
    uint64_t epoch;
    uint32_t fshift;
};
struct OvwTables {
    OvwEv prev, cur;           // read / written by the round being launched
    uint32_t* filt_next;       // the filter the NEXT round writes: cleared by this one
    uint32_t filt_words;
    uint4* log;                // the pieces' logs, 2 x (lk positions of the piece) entries from 2 x kt.piece_base[piece]; a round rewrites a log in
                               // place, entry by entry behind its reading of the old one
    uint32_t* res;             // 8 words per listed piece: entries, NbProcessed, NbSkipped, NbJCheckKmer, NbNoJuncs, tests run on the spot, late true test
    // change marks, by hash of the k-mer: the earliest piece (time) whose entry at that k-mer posts other events in this round than in the round
    // before.  A piece none of whose registered k-mers carries a mark of an EARLIER piece reads in the next round what it read in this one: it
    // keeps its log and only posts its events again (the event tables are rebuilt every round).
    const uint32_t* mark_prev; // written by the round before
    uint32_t* mark_cur;        // written by this round
    uint32_t* mark_next;       // cleared by this round
    uint32_t mark_mask;
    uint32_t* list;            // window-local indices of the pieces walked this way (k_ko_prepare)
    uint32_t list_cap;
    uint32_t* state;           // [0] pieces listed, [1] failed (overflow), [2] rounds run, [3] some cluster holds a long piece, [5] blocks of k_ovw_commit done,
                               // [6], [7] entries of the window's rounds found in the two event tables, [8 + r] some log changed in round r
    uint32_t* longp;           // per root: the cluster holds a piece of more than 128 windows
};

// the round whose logs are the settled ones: the first round >= 1 in which no log changed; -1: none (yet), or the tables overflowed
__device__ __forceinline__ int ovw_settled_round(const uint32_t* state, int rounds) {
    if (state[1]) return -1;
    for (int r = 1; r < rounds; r++)
        if (state[8 + r] == 0) return r;
    return -1;
}

// ---- key-ordered walk: preparation and the walk itself (see KoTables) ----------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_ko_reset(KoTables kt, uint32_t n_pieces, uint32_t parity) {
    // the tables are only touched by windows that hold a large cluster (state[4 + parity of that window]): most windows find them clean
    const bool dirty = kt.state[4 + (parity ^ 1u)] != 0;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    if (dirty) {
        for (uint32_t a = i; a <= kt.hk_mask; a += stride) { kt.hk_key[a] = KO_EMPTY; kt.hk_head[a] = U_INF; kt.hk_turn[a] = 0; }
        for (uint32_t a = i; a < n_pieces; a += stride) kt.bad[a] = 0;
    }
    if (i < 4) kt.state[i] = 0;
    if (i == 4) kt.state[4 + parity] = 0;
}

// one thread per piece of the window: the pieces of large clusters list their lk positions as occurrences of their k-mers
__global__ void __launch_bounds__(256) k_ko_prepare(Planes pl, const uint32_t* __restrict__ root, const uint32_t* __restrict__ count,
                                                    const WinDesc* __restrict__ wdp, KoTables kt, uint32_t heavy, uint32_t parity,
                                                    const uint32_t* __restrict__ weight, uint32_t heavy_w, OvwTables ot) {
    const WinDesc wd = *wdp;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= wd.n) return;
    const uint32_t r = root[i];
    if (!ko_cluster(count[r], weight[r], heavy, heavy_w)) return;
    kt.state[4 + parity] = 1u;                                        // the tables are in use: the next window resets them
    atomicAdd(&kt.state[3], 1u);
    const uint2 pc = pl.pieces[wd.first_piece + i];
    if (pc.y > 64 * KO_CHUNKS) { kt.bad[r] = 1u; return; }           // the turn bookkeeping keeps KO_CHUNKS words of positions per piece
    if (ot.list) {                                                    // the optimistic walk takes the clusters whose pieces all fit its register view
        if (pc.y > 128) { ot.longp[r] = 1u; ot.state[3] = 1u; }
        else {
            const uint32_t at = atomicAdd(&ot.state[0], 1u);
            if (at < ot.list_cap) ot.list[at] = i; else ot.state[1] = 1u;
        }
    }
    uint32_t n = 0;
    for (uint32_t c = 0; c < KO_CHUNKS && c * 64 < pc.y; c++) n += (uint32_t)__popcll(fd_bits_at(pl.lk, pc.x + 64 * c) & chunk_mask(pc.y, c));
    const uint32_t base = atomicAdd(&kt.state[0], n);
    if (base + n > kt.occ_cap) { atomicOr(&kt.state[1], 1u); return; }
    kt.piece_base[i] = base;
    uint32_t node = base;
    for (uint32_t c = 0; c < KO_CHUNKS && c * 64 < pc.y; c++) {
        uint64_t w = fd_bits_at(pl.lk, pc.x + 64 * c) & chunk_mask(pc.y, c);
        while (w) {
            const uint32_t q = 64 * c + (uint32_t)__builtin_ctzll(w);
            w &= w - 1;
            uint32_t h = pl.kh[pc.x + q];
            if (h == KO_EMPTY) h = KO_EMPTY - 1;
            uint32_t s = (h * 0x9E3779B1u) & kt.hk_mask, e = U_INF;
            for (uint32_t probe = 0; probe < 4096; probe++) {
                const uint32_t old = atomicCAS(&kt.hk_key[s], KO_EMPTY, h);
                if (old == KO_EMPTY || old == h) { e = s; break; }
                s = (s + 1) & kt.hk_mask;
            }
            if (e == U_INF) { atomicOr(&kt.state[1], 1u); return; }
            kt.occ_entry[node] = e;
            kt.occ_id[node] = ((uint64_t)i << 32) | q;
            kt.occ_next[node] = atomicExch(&kt.hk_head[e], node);
            node++;
        }
    }
}

// rank of every occurrence among the occurrences of its k-mer, by (piece, position)
__global__ void __launch_bounds__(256) k_ko_rank(KoTables kt, const uint32_t* __restrict__ ovw_state, int ovw_rounds) {
    if (kt.state[1] & 1u) return;
    if (ovw_state && !ovw_state[3] && ovw_settled_round(ovw_state, ovw_rounds) >= 0) return;   // the optimistic walk has left nothing for the turns to order
    const uint32_t n = kt.state[0];
    for (uint32_t node = blockIdx.x * blockDim.x + threadIdx.x; node < n; node += gridDim.x * blockDim.x) {
        const uint64_t mine = kt.occ_id[node];
        uint32_t r = 0;
        for (uint32_t m = kt.hk_head[kt.occ_entry[node]]; m != U_INF; m = kt.occ_next[m]) r += kt.occ_id[m] < mine ? 1u : 0u;
        kt.occ_rank[node] = r;
    }
}

// Round 3 rebuilt this walk three ways and measured all of them SLOWER than the form below (twenty-copy repeat set, same box: 522 ms here;
// 565 ms with the piece's occurrences prefetched into LDS, key word + both records fetched in one round trip and reused for the second
// facing, the next turn counter requested a visit ahead and turns stored lazily; 631 ms with all of that on the scalar unit -- the whole wave
// walking with wave-uniform values; 663 ms confined to one XCD with L2-resident hand-overs).  The walk is bound by the issue rate of ONE wave's
// instruction stream (~1 000-2 000 instructions per junction visit at one instruction per 4-5 cycles), so every scheme that saves memory
// round trips by adding bookkeeping loses, and the scalar unit loses to register pressure (the walk's state is ~200 SGPRs' worth against
// the 100 a wave has: 3 774 spill moves in the kernel).  The experimental code is in the history (commit 530a5c9); the numbers and the
// per-piece trace are in profiles/r03_ko_walk.txt, the reading in DESIGN.md section 4.1.
// The walk: waves draw chunks of 64 consecutive pieces from a ticket counter (whoever holds a ticket is running, and every piece a wave can
// wait for belongs to the same or an earlier ticket); lane 0 walks the chunk's pieces of large clusters one after the other.
__global__ void __launch_bounds__(64) k_walk_ko(Planes pl, FdParams fp, JTable jt, const uint32_t* __restrict__ root, const uint32_t* __restrict__ count,
                                                const WinDesc* __restrict__ wdp, uint64_t piece_seq_base, const uint32_t* __restrict__ bloom, DevCounters* cnt,
                                                KoTables kt, uint32_t heavy, uint32_t KO_TICKET, const uint32_t* __restrict__ weight, uint32_t heavy_w,
                                                const uint32_t* __restrict__ ovw_state, const uint32_t* __restrict__ ovw_longp, int ovw_rounds) {
    const WinDesc wd = *wdp;
    if (kt.state[1] & 1u) return;                                      // a table overflowed: k_walk takes every cluster
    // the optimistic walk has settled and applied the clusters without long pieces: only the others are left
    const bool ovw_done = ovw_state && ovw_settled_round(ovw_state, ovw_rounds) >= 0;
    if (ovw_done && !ovw_state[3]) return;
    WalkCtx wc;
    KoState& ko = wc.ko;
    wc.pl = pl; wc.fp = fp; wc.jt = jt; wc.cnt = cnt; wc.bloom = bloom; wc.win_seq = 0;
    wc.nb_processed = wc.nb_skipped = wc.nb_jcheck = wc.nb_no_juncs = wc.n_created = wc.n_filled = 0;
    wc.created_now = false; wc.fail = 0; wc.dbg = 0;
    ko.kt = kt;
    ko.cnt = cnt;
    unsigned long long walked = 0;
    if (kt.state[3] == 0) return;                                      // no large cluster in this window (the usual case)
#ifdef FGPU_KO_ONE_XCD
    // experiment: only the waves that landed on XCD 0 take tickets (the grid is eight times as large), so that every hand-over stays
    // inside one XCD; the accesses keep their agent scope, only their latency is looked at
    if ((__builtin_amdgcn_s_getreg(6164) & 15) != 0) return;           // hwreg(HW_REG_XCC_ID, 0, 4)
#endif
    // Pieces per ticket.  KO_TICKET (a multiple of 64) where the pieces of large clusters are few among the window's pieces -- a
    // same-address atomic per 64 pieces cost 0.2 ms per window -- so that a ticket holds about one of them.  Where they are dense (a small
    // genome at high coverage: EVERY piece of the reference's own 1 000-read example is in one cluster) a ticket of 64 would put 64
    // consecutive pieces of the cluster on one lane, one after the other, and the waves would follow each other through the window: the
    // whole walk one serial chain (33 ms for those 1 000 reads, twice what the cluster's single thread in k_walk needs).  So the ticket
    // shrinks with the density, down to one piece: the pieces of a cluster then spread over the waves and overlap as their k-mers allow.
    uint32_t span = KO_TICKET;
    {
        const uint32_t per = wd.n / kt.state[3];          // window pieces per piece of a large cluster (state[3] != 0 here)
        while (span > 1 && span > per) span >>= 1;
    }
    for (;;) {
        uint32_t ticket = 0;
        if (fd_lane() == 0) ticket = atomicAdd(&kt.state[2], 1u);
        ticket = (uint32_t)__shfl((int)ticket, 0, 64);
        if ((uint64_t)ticket * span >= wd.n) break;
        for (uint32_t sub = 0; sub < (span + 63) / 64; sub++) {
            const uint32_t first = ticket * span + sub * 64;
            if (first >= wd.n) break;
            const uint32_t i = first + (uint32_t)fd_lane();
            bool mine = false;
            if (i < wd.n && (uint32_t)fd_lane() < span) {
                const uint32_t r = root[i];
                mine = ko_cluster(count[r], weight[r], heavy, heavy_w) && !kt.bad[r] && !(ovw_done && !ovw_longp[r]);
                if (mine && r == i) {                     // the statistics k_walk keeps per cluster
                    atomicAdd(&cnt->followers, (unsigned long long)count[r]);
                    atomicMax(&cnt->max_cluster, (unsigned long long)count[r] + 1);
                }
            }
            uint64_t todo = __ballot(mine);
            if (fd_lane() == 0) {
                while (todo) {
                    const uint32_t b = (uint32_t)__builtin_ctzll(todo);
                    todo &= todo - 1;
                    const uint32_t li = first + b;
                    const uint2 pc = pl.pieces[wd.first_piece + li];
                    ko.base = kt.piece_base[li];
                    ko.done = 0;
                    ko.mid = (pc.y + (uint32_t)fp.k - 1) / 2 - (uint32_t)fp.k / 2;
                    ko.last.n = ko.cur.n = ko.fake.n = 0;
                    ko.cur_q = 0;
                    ko.cur_in = 0;
                    ko.wait_acc = 0;
                    ko.stamp_base = nullptr;
                    ko.stamp_n = 0;
                    uint32_t n_lk = 0;
                    for (uint32_t c = 0; c < KO_CHUNKS; c++) {
                        ko_lk(c) = c * 64 < pc.y ? fd_bits_at(pl.lk, pc.x + 64 * c) & chunk_mask(pc.y, c) : 0ULL;
                        n_lk += (uint32_t)__popcll(ko_lk(c));
                        ko_absent(true, c) = 0;
                        ko_absent(false, c) = 0;
                    }
#ifdef FGPU_KO_TRACE
                    const unsigned long long trace_t0 = wall_clock64();
                    ko.stamp_t0 = trace_t0;
                    if (kt.trace && ((piece_seq_base + wd.first_piece + li) & 15) == 0) {
                        const unsigned long long slot = atomicAdd(&kt.trace[1], 1ULL);
                        if (slot < 4096) ko.stamp_base = kt.trace + (1ULL << 23) + slot * 1024;
                    }
#endif
#ifdef FGPU_KO_TIMING
                    const unsigned long long tp = wall_clock64();
#endif
                    walk_piece<WALK_KO>(wc, pc.x, pc.y, piece_seq_base + wd.first_piece + li);
                    ko_finish(wc, pc.y);
                    walked++;
#ifdef FGPU_KO_TRACE
                    if (kt.trace) {           // one record per walked piece: number, start, end, ticks waited | lk positions << 48
                        const unsigned long long at = atomicAdd(&kt.trace[0], 1ULL);
                        if (at < (1ULL << 21)) {
                            unsigned long long* rec = kt.trace + 4 + 4 * at;
                            rec[0] = piece_seq_base + wd.first_piece + li;
                            rec[1] = trace_t0;
                            rec[2] = wall_clock64();
                            rec[3] = ko.wait_acc | ((unsigned long long)n_lk << 48);
                        }
                    }
                    if (ko.stamp_base) ko.stamp_base[0] = ((piece_seq_base + wd.first_piece + li) << 16) | (ko.stamp_n < 1020 ? ko.stamp_n : 1020);
#else
                    (void)n_lk;
#endif
#ifdef FGPU_KO_TIMING
                    atomicAdd(&cnt->par_probe[0], wall_clock64() - tp);
                    atomicAdd(&cnt->par_probe[2], 1ULL);
#endif
                }
            }
        }
    }
    if (fd_lane() == 0) {
        if (wc.nb_processed) atomicAdd(&cnt->nb_processed, wc.nb_processed);
        if (wc.nb_skipped) atomicAdd(&cnt->nb_skipped, wc.nb_skipped);
        if (wc.nb_jcheck) atomicAdd(&cnt->nb_jcheck, wc.nb_jcheck);
        if (wc.nb_no_juncs) atomicAdd(&cnt->nb_no_juncs, wc.nb_no_juncs);
        if (wc.n_created) atomicAdd(&cnt->n_junctions, wc.n_created);
        if (wc.n_filled) atomicAdd(&cnt->flags_filled, wc.n_filled);
        if (walked) atomicAdd(&cnt->walk_parallel, walked);
    }
}

// ---- the optimistic walk of large clusters (k_ovw_round, k_ovw_commit) ------------------------------------------------------------------------
// Round 4.  The key-ordered walk above keeps the sequential order access by access, and what it pays for that is one wave's instruction stream
// per piece and a chain of hand-overs: ~300 us per repeat piece, ~4 pieces at a time (profiles/r03_ko_walk.txt).  But the walk of a piece is a
// FUNCTION of what it reads in the junction map, and it reads little: whether the keys at its registered positions exist, and one distance per
// visit (its skip).  What it writes splits into what other pieces' paths can depend on -- creations and raised distances -- and what they
// cannot (coverage counts, link flags: order-free).  So the pieces of a large cluster are walked ALL AT ONCE, read-only against the map as the
// window found it, every piece noting what it WOULD write in a private log and posting the path-relevant part as EVENTS with its file-order
// time (the index of the piece in the window):
//      (record, distance index) -> (time, value)         a distance raised above the stored one
//      (record, creation)       -> time                  a key that is not in the map yet
// and reading, besides the stored record, the events of EARLIER pieces as the previous round left them:
//      exists at time t   <=>  present in the map  or  a creation event with time < t  (or created by the piece itself)
//      distance at time t  =   max(stored, events with time < t, the piece's own earlier contributions).
// Round after round every piece walks again on the previous round's events, until no piece's log differs from its log of the round before.
// That fixed point is the sequential walk: the earliest piece of the cluster reads no events at all, so its log is final after round 0; a piece
// whose predecessors' logs are final reads exactly what the sequential run shows it, so its log is final one round later -- induction over the
// file order, the argument of pass 1's first-set times and of the long pair filter (pairs.hip).  Rounds needed = the longest
// chain of pieces that really hand information on (creations while a repeat is first met; two rounds once its records stand), not the number
// of pieces.  The settled logs are then applied with atomics -- coverage saturating, distances by maximum, link flags by OR, each creation by
// its one creator -- and the counters of the settled paths are added up.  Nothing of a round that did not settle reaches the map except claimed
// slots, which mean nothing without their presence bits.
// Pieces of up to 128 windows (the register view); a cluster that holds a longer one, rounds that do not settle within OVW_MAX_ROUNDS, or tables
// that overflow leave the cluster to the key-ordered walk, which runs behind and looks at the outcome first.
__device__ __forceinline__ uint32_t ovw_filter_bit(const OvwEv& ev, uint64_t k) { return ((uint32_t)fd_mix(k) * 0x85EBCA6Bu) >> ev.fshift; }
__device__ __forceinline__ unsigned long long ld_agent_ull(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// post (record, index) -> (time, value).  An event is DOMINATED by one of the same key that is not later and not smaller: it can never decide a
// reader's maximum, so it is dropped (or replaced where it dominates) -- the entries of a key stay a short staircase however many pieces of a
// repeat post the same distance.  false: table full.
__device__ __forceinline__ bool ovw_post(const OvwEv& ev, uint32_t rec, uint32_t idx, uint32_t time, uint32_t value) {
    const uint64_t k = ((uint64_t)rec << 3) | idx;
    const unsigned long long kw_mine = ev.epoch | k, vw_mine = ev.epoch | ((unsigned long long)time << 8) | value;
    uint64_t s = fd_mix(k) & ev.mask;
    const uint64_t limit = ev.mask < 4096 ? ev.mask : 4096;
    for (uint64_t n = 0; n <= limit; n++) {
        unsigned long long kw = ld_agent_ull(&ev.key[s]);
        if ((kw & OVW_EPOCH_MASK) != ev.epoch) {                      // free: claim it
            const unsigned long long old = atomicCAS(&ev.key[s], kw, kw_mine);
            if (old == kw) {
                __hip_atomic_store(&ev.val[s], vw_mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t b = ovw_filter_bit(ev, k);
                atomicOr(&ev.bits[b >> 5], 1u << (b & 31));
                return true;
            }
            kw = old;
        }
        if (kw == kw_mine) {
            for (;;) {
                const unsigned long long vw = ld_agent_ull(&ev.val[s]);
                if ((vw & OVW_EPOCH_MASK) != ev.epoch) break;          // claimed, value not stored yet: nothing can be concluded from it
                const uint32_t t2 = (uint32_t)(vw >> 8), v2 = (uint32_t)(vw & 0xFF);
                if (t2 <= time && v2 >= value) return true;            // dominated
                if (!(time <= t2 && value >= v2)) break;               // neither dominates: both stay
                if (atomicCAS(&ev.val[s], vw, vw_mine) == vw) return true;
            }
        }
        s = (s + 1) & ev.mask;
    }
    return false;
}
// max value over the events of (record, index) with a time before `time` (0: none)
__device__ __forceinline__ uint32_t ovw_before(const OvwEv& ev, uint32_t rec, uint32_t idx, uint32_t time) {
    const uint64_t k = ((uint64_t)rec << 3) | idx;
    const uint32_t b = ovw_filter_bit(ev, k);
    if (!((ev.bits[b >> 5] >> (b & 31)) & 1u)) return 0;
    const unsigned long long kw_mine = ev.epoch | k;
    uint64_t s = fd_mix(k) & ev.mask;
    uint32_t best = 0;
    for (uint64_t n = 0; n <= ev.mask; n++) {
        const unsigned long long kw = ev.key[s];
        if ((kw & OVW_EPOCH_MASK) != ev.epoch) break;
        if (kw == kw_mine) {
            const unsigned long long vw = ev.val[s];
            const uint32_t t2 = (uint32_t)(vw >> 8), v2 = (uint32_t)(vw & 0xFF);
            if (t2 < time && v2 > best) best = v2;
        }
        s = (s + 1) & ev.mask;
    }
    return best;
}

struct OvwWalk {
    OvwEv prev, cur;
    bool have_prev;
    uint4* log;                // rewritten in place: entry i of the old log is read before entry i of the new one is written
    bool have_old;
    uint32_t n_old, n_new, cap;
    uint32_t time;
    bool changed, overflow;
    const uint32_t* kh;        // the batch's k-mer hashes (Planes::kh), for the change marks
    uint64_t p0;
    uint32_t* mark;
    uint32_t mark_mask;
};
// Which records does the piece have an entry for already?  A filter of 1024 bits per lane in LDS, two bits per record: a hit sends the visit
// through the piece's own log (the same junction twice on a piece: tandem repeats, twice in a row) -- with the 64 bits of a register a piece
// inside a repeat, a hundred records, went through its log at every visit (10^4 log reads per piece and round).
constexpr uint32_t OVW_SEEN_WORDS = 32;
__device__ __forceinline__ uint32_t* ovw_seen_words() {
    __shared__ uint32_t s_seen[OVW_SEEN_WORDS * 64];
    return s_seen + (threadIdx.x & 63);
}
__device__ __forceinline__ void ovw_seen_clear() {
    uint32_t* w = ovw_seen_words();
    for (uint32_t i = 0; i < OVW_SEEN_WORDS; i++) w[i * 64] = 0;
}
__device__ __forceinline__ bool ovw_seen_test_and_set(uint32_t rec) {
    const uint64_t h = fd_mix(rec);
    const uint32_t a = (uint32_t)h & (OVW_SEEN_WORDS * 32 - 1), b = (uint32_t)(h >> 32) & (OVW_SEEN_WORDS * 32 - 1);
    uint32_t* w = ovw_seen_words();
    const uint32_t wa = w[(a >> 5) * 64], wb = w[(b >> 5) * 64];
    const bool hit = ((wa >> (a & 31)) & 1u) && ((wb >> (b & 31)) & 1u);
    w[(a >> 5) * 64] = wa | (1u << (a & 31));
    w[(b >> 5) * 64] |= 1u << (b & 31);
    return hit;
}
// entry: record | half-step, flags, backward distance | forward distance, the STORED record's dist[4] | its dist[0..3] -- the stored bytes ride
// along so that the next round's visit of the same half-step needs no table access at all (the map does not change between rounds)
__device__ __forceinline__ uint4 ovw_pack(uint32_t rec, uint32_t tn, uint32_t flags, uint32_t d_bwd, uint32_t d_fwd, uint64_t stored_lo) {
    return make_uint4(rec, tn | (flags << 16) | (d_bwd << 24), d_fwd | ((uint32_t)((stored_lo >> 32) & 0xFF) << 8), (uint32_t)stored_lo);
}
// what an entry posts: creation, backward distance, forward distance -- each only where it exceeds what is stored (the stored bytes ride in the
// entry); 0 = the entry posts nothing
__device__ __forceinline__ uint32_t ovw_event_sig(const uint4& e) {
    const uint32_t tn = e.y & 0xFFFF, flags = (e.y >> 16) & 0xFF, real = (flags & OVW_F_REAL) >> 1;
    const bool fwd = tn & 1;
    const uint32_t ib = fwd ? 4u : real, iff = fwd ? real : 4u;
    const uint64_t stored = (uint64_t)e.w | ((uint64_t)((e.z >> 8) & 0xFF) << 32);
    const uint32_t d_bwd = e.y >> 24, d_fwd = e.z & 0xFF;
    uint32_t sig = (flags & OVW_F_CREATED) ? 1u : 0u;
    if (d_bwd > ((stored >> (8 * ib)) & 0xFF)) sig |= 2u | (d_bwd << 8);
    if ((flags & OVW_F_DF) && d_fwd > ((stored >> (8 * iff)) & 0xFF)) sig |= 4u | (d_fwd << 16);
    return sig;
}
__device__ __forceinline__ void ovw_mark(const OvwWalk& ow, const uint4& e) {
    const uint32_t h = ow.kh[ow.p0 + ((e.y & 0xFFFF) >> 1)];
    atomicMin(&ow.mark[(h * 0x9E3779B1u) >> 8 & ow.mark_mask], ow.time);
}
__device__ __forceinline__ void ovw_emit(OvwWalk& ow, const uint4& e) {
    if (ow.n_new >= ow.cap) { ow.overflow = true; return; }
    const uint32_t sig = ovw_event_sig(e);
    if (!ow.have_old || ow.n_new >= ow.n_old) {
        ow.changed = true;
        if (sig) ovw_mark(ow, e);
    } else {
        const uint4 o = ow.log[ow.n_new];
        if (o.x != e.x || o.y != e.y || o.z != e.z) ow.changed = true;
        const uint32_t osig = ovw_event_sig(o);
        if (o.x != e.x || osig != sig) {           // the events of this entry are not those of the round before: whoever reads them walks again
            if (sig) ovw_mark(ow, e);
            if (osig) ovw_mark(ow, o);
        }
    }
    ow.log[ow.n_new++] = e;
}
// the events of a log entry, posted again (a piece that keeps its log: the tables are rebuilt every round)
__device__ __forceinline__ bool ovw_repost(const OvwEv& cur, const uint4& e, uint32_t time) {
    const uint32_t sig = ovw_event_sig(e);
    if (!sig) return true;
    const uint32_t tn = e.y & 0xFFFF, flags = (e.y >> 16) & 0xFF, real = (flags & OVW_F_REAL) >> 1;
    const bool fwd = tn & 1;
    const uint32_t ib = fwd ? 4u : real, iff = fwd ? real : 4u;
    bool ok = true;
    if (sig & 1u) ok = ok && ovw_post(cur, e.x, OVW_CREATION, time, 1);
    if (sig & 2u) ok = ok && ovw_post(cur, e.x, ib, time, e.y >> 24);
    if (sig & 4u) ok = ok && ovw_post(cur, e.x, iff, time, e.z & 0xFF);
    return ok;
}
// the piece's own earlier contributions to (record, index) -- the same junction twice on a piece: tandem repeats, or twice in a row
__device__ __forceinline__ uint32_t ovw_own(const OvwWalk& ow, uint32_t rec, uint32_t idx, const uint4& pending, bool have_pending) {
    uint32_t best = 0;
    for (uint32_t i = 0; i <= ow.n_new; i++) {
        if (i == ow.n_new && !have_pending) break;
        const uint4 e = i == ow.n_new ? pending : ow.log[i];
        if (e.x != rec) continue;
        const uint32_t tn = e.y & 0xFFFF, flags = (e.y >> 16) & 0xFF, real = (flags & OVW_F_REAL) >> 1;
        const bool fwd = tn & 1;
        const uint32_t ib = fwd ? 4u : real, iff = fwd ? real : 4u;
        if (ib == idx) best = max(best, e.y >> 24);
        if (iff == idx && (flags & OVW_F_DF)) best = max(best, e.z & 0xFF);
    }
    return best;
}
__device__ __forceinline__ bool ovw_own_created(const OvwWalk& ow, uint32_t rec) {
    for (uint32_t i = 0; i < ow.n_new; i++) {
        const uint4 e = ow.log[i];
        if (e.x == rec && ((e.y >> 16) & OVW_F_CREATED)) return true;
    }
    return false;
}

// What the snapshot planes cannot know at the piece's registered positions (see created_bits): the live table, plus the creations earlier
// pieces posted in the previous round.
__device__ __noinline__ uint4 ovw_live_bits(const uint64_t* __restrict__ codes, int k, uint64_t p, uint64_t where, uint64_t* jkeys, uint64_t jmask,
                                            const OvwEv* prev, uint32_t time) {
    JTable jt;
    jt.keys = jkeys;
    jt.mask = jmask;
    jt.recs = nullptr; jt.stamps = nullptr; jt.filter = nullptr; jt.filter_mask = 0;
    uint64_t mF = 0, mB = 0;
    while (where) {
        const uint32_t i = (uint32_t)__builtin_ctzll(where);
        where &= where - 1;
        const uint64_t km = fd_kmer_at(codes, p + i, k);
        const uint64_t rc = fd_revcomp(km, k);
        const uint64_t canon = km < rc ? km : rc;
        uint64_t slot;
        uint32_t present;
        if (jt_find_live(jt, canon, slot, present)) {
            const uint32_t oF = km == canon ? 0u : 1u, oB = rc == canon ? 0u : 1u;
            bool f = (present >> oF) & 1u, b = (present >> oB) & 1u;
            if (prev) {
                if (!f) f = ovw_before(*prev, (uint32_t)(slot * 2 + oF), OVW_CREATION, time) != 0;
                if (!b) b = oB == oF ? f : ovw_before(*prev, (uint32_t)(slot * 2 + oB), OVW_CREATION, time) != 0;
            }
            if (f) mF |= 1ULL << i;
            if (b) mB |= 1ULL << i;
        }
    }
    return make_uint4((uint32_t)mF, (uint32_t)(mF >> 32), (uint32_t)mB, (uint32_t)(mB >> 32));
}
__device__ __forceinline__ void ovw_created_bits(const WalkCtx& wc, const OvwWalk& ow, const PieceView& v, uint32_t c, uint64_t& mF, uint64_t& mB) {
    const uint64_t lk = c == 0 ? v.lk0 : v.lk1, inF = c == 0 ? v.inF0 : v.inF1, inB = c == 0 ? v.inB0 : v.inB1;
    const uint64_t where = lk & ~(inF & inB);
    if (!where) { mF = mB = 0; return; }
    const uint4 r = ovw_live_bits(wc.pl.codes, wc.fp.k, v.p0 + 64 * c, where, wc.jt.keys, wc.jt.mask, ow.have_prev ? &ow.prev : nullptr, ow.time);
    mF = (uint64_t)r.x | ((uint64_t)r.y << 32);
    mB = (uint64_t)r.z | ((uint64_t)r.w << 32);
}
// the piece has just created the record keyed by `key`: the same k-mer further along the piece (tandem repeats) is in the map from now on
__device__ __forceinline__ void ovw_mark_created(const WalkCtx& wc, PieceView& v, uint64_t key) {
    const int k = wc.fp.k;
    for (uint32_t c = 0; c < 2; c++) {
        uint64_t w = c == 0 ? v.lk0 : v.lk1;
        while (w) {
            const uint32_t b = (uint32_t)__builtin_ctzll(w);
            w &= w - 1;
            const uint64_t km = pv_kmer(v, wc.pl.codes, v.p0 + 64 * c + b, k);
            if (km == key) { if (c == 0) v.xF0 |= 1ULL << b; else v.xF1 |= 1ULL << b; }
            if (fd_revcomp(km, k) == key) { if (c == 0) v.xB0 |= 1ULL << b; else v.xB1 |= 1ULL << b; }
        }
    }
}

// one junction visit (or add_fake_junction's): the record is found (a key that has no slot yet gets one), its stored distances are read, the
// contribution of this visit to `idx_b` is posted if it raises the stored value.  Returns false on a full table.
struct OvwRec {
    uint32_t rec;
    uint64_t lo;       // the stored record's low word (distances in bytes 0-4); zero for a record that is not in the map
    bool present;
};
__device__ __forceinline__ bool ovw_record(WalkCtx& wc, uint64_t key, OvwRec& out) {
    const uint64_t rc = fd_revcomp(key, wc.fp.k);
    const uint64_t canon = key < rc ? key : rc;
    const int orient = key == canon ? 0 : 1;
    const uint64_t home = fd_mix(canon) & wc.jt.mask;
    const uint64_t* spec = (const uint64_t*)(wc.jt.recs + (home * 2 + orient) * 16);
    const uint64_t w_first = ld_agent(&wc.jt.keys[home]);
    const uint64_t spec_lo = spec[0];
    uint64_t slot;
    uint32_t present;
    if (!jt_find_or_claim(wc.jt, canon, home, w_first, slot, present, wc.cnt)) return false;
    out.rec = (uint32_t)(slot * 2 + orient);
    out.present = (present >> orient) & 1u;
    out.lo = !out.present ? 0ULL : slot == home ? spec_lo : ((const uint64_t*)(wc.jt.recs + (slot * 2 + orient) * 16))[0];
    return true;
}

// scan_forward (ReadScanner.cpp:112-206) for the piece {p0, nwin <= 128}, read-only: see the section's header.  wc.fail: 1 table full, 4 a junction
// test the preview had left out came out true (the lazy scan is void if this path settles).
__device__ __forceinline__ void ovw_walk_piece(WalkCtx& wc, OvwWalk& ow, uint64_t p0, uint32_t nwin) {
    const int k = wc.fp.k, j = wc.fp.j;
    const int tmax = 2 * (int)nwin - 2 - 2 * j;
    const int spacer = 2 * wc.fp.max_spacer - 1;
    PieceView v;
    pv_load(v, wc.pl, p0, nwin);
    ovw_created_bits(wc, ow, v, 0, v.xF0, v.xB0);
    if (nwin > 64) ovw_created_bits(wc, ow, v, 1, v.xF1, v.xB1);
    int t = 2 * j + 1;
    int last_pos = 0;
    bool have_last = false;
    uint4 pend = make_uint4(0, 0, 0, 0);            // the last visit's entry: its forward distance is known at the next visit (or at the end)
    uint64_t last_lo = 0;                           // stored distances of the last visit's record
    int last_t = 0, last_ext_fwd = 0;

    while (t <= tmax) {
        int t_sp = last_pos + spacer;
        if (t_sp < t) t_sp = t;
        int tn;
        uint32_t q = 0;
        bool fwd = false, in_map = false, by_spacer = false;
        for (;;) {
            int t_ev = 0x7fffffff;
            bool ev_in_map = false;
            {
                const uint32_t q0 = (uint32_t)(t >> 1);
                const int t_stop = tmax < t_sp ? tmax : t_sp;
                for (uint32_t c = q0 >> 6; c * 64 < nwin && 2 * (int)(c * 64) <= t_stop; c++) {
                    const uint64_t mF = c == 0 ? (v.inF0 | v.xF0) : (v.inF1 | v.xF1), mB = c == 0 ? (v.inB0 | v.xB0) : (v.inB1 | v.xB1);
                    uint64_t eF = mF | (c == 0 ? v.fF0 : v.fF1);
                    uint64_t eB = mB | (c == 0 ? v.fB0 : v.fB1);
                    if (c == (q0 >> 6)) {
                        const uint64_t from = ~0ULL << (q0 & 63);
                        eF &= from;
                        eB &= from;
                        if (t & 1) eB &= ~(1ULL << (q0 & 63));
                    }
                    const int tb = eB ? 2 * (int)(c * 64 + __builtin_ctzll(eB)) : 0x7fffffff;
                    const int tf = eF ? 2 * (int)(c * 64 + __builtin_ctzll(eF)) + 1 : 0x7fffffff;
                    const int te = tb < tf ? tb : tf;
                    if (te != 0x7fffffff) {
                        t_ev = te;
                        ev_in_map = ((te & 1) ? mF : mB) >> ((te >> 1) & 63) & 1ULL;
                        break;
                    }
                }
            }
            tn = t_ev < t_sp ? t_ev : t_sp;
            if (tn > tmax) {
                if (fill_missing<WALK_OVW>(wc, v, t, tmax + 1)) continue;
                break;
            }
            q = (uint32_t)(tn >> 1);
            fwd = tn & 1;
            if (tn == t_ev) {
                in_map = ev_in_map;
            } else {
                const uint32_t c = q >> 6;
                const uint64_t mF = c == 0 ? (v.inF0 | v.xF0) : (v.inF1 | v.xF1), mB = c == 0 ? (v.inB0 | v.xB0) : (v.inB1 | v.xB1);
                in_map = ((fwd ? mF : mB) >> (q & 63)) & 1ULL;
            }
            by_spacer = !in_map && (tn - last_pos >= spacer);
            if (fill_missing<WALK_OVW>(wc, v, t, (in_map || by_spacer) ? tn : tn + 1)) continue;
            break;
        }
        if (tn > tmax) {
            wc.nb_processed += (unsigned long long)(tmax - t + 1);
            wc.nb_jcheck += jcheck_sum(wc, v, t, tmax + 1);
            break;
        }
        wc.nb_processed += (unsigned long long)(tn - t);
        wc.nb_jcheck += jcheck_sum(wc, v, t, (in_map || by_spacer) ? tn : tn + 1);

        // ---- junction at (q, fwd)  (ReadScanner.cpp:133-192)
        const uint64_t km = pv_kmer(v, wc.pl.codes, p0 + q, k);
        const uint64_t key = fwd ? km : fd_revcomp(km, k);
        const int real = fwd ? pv_base(v, wc.pl.codes, p0 + q + k) : (pv_base(v, wc.pl.codes, p0 + q - 1) ^ 2);
        const int ext_fwd = fwd ? real : 4, ext_bwd = fwd ? 4 : real;
        OvwRec r;
        {   // the previous round's visit of this very half-step knows the record and what is stored in it
            const uint32_t at = ow.n_new + (have_last ? 1u : 0u);          // (the last visit's entry is still pending: this one follows it)
            const uint4 o = ow.have_old && at < ow.n_old ? ow.log[at] : make_uint4(0, 0xFFFFFFFFu, 0, 0);
            if ((o.y & 0xFFFF) == (uint32_t)tn && !((o.y >> 16) & OVW_F_FAKE)) {
                r.rec = o.x;
                r.present = ((o.y >> 16) & OVW_F_PRESENT) != 0;
                r.lo = (uint64_t)o.w | ((uint64_t)((o.z >> 8) & 0xFF) << 32);
            } else if (!ovw_record(wc, key, r)) { wc.fail = 1; return; }
        }
        const bool revisit = ovw_seen_test_and_set(r.rec);
        last_pos = tn;
        const uint32_t d_in = (uint32_t)((have_last ? tn - last_t : tn - 2 * j) & 0xFF);    // Junction::update narrows to a byte
        if (have_last) {       // directLinkJunctions (JunctionMap.cpp:551-561): the last junction's forward distance, this one's backward distance
            pend.y |= (OVW_F_LINK_F | OVW_F_DF) << 16;
            pend.z = (pend.z & ~0xFFu) | d_in;
            if (d_in > ((last_lo >> (8 * last_ext_fwd)) & 0xFF) && !ovw_post(ow.cur, pend.x, (uint32_t)last_ext_fwd, ow.time, d_in)) { wc.fail = 1; return; }
            ovw_emit(ow, pend);    // (before the log is asked about this record: the last visit may have been to the same junction)
        }
        // createJunction (JunctionMap.cpp:567-570): the key is neither stored, nor created by an earlier piece, nor by this piece earlier on
        const bool created = !r.present && !(ow.have_prev && ovw_before(ow.prev, r.rec, OVW_CREATION, ow.time)) && !(revisit && ovw_own_created(ow, r.rec));
        uint32_t flags = (created ? OVW_F_CREATED : 0u) | ((uint32_t)real << 1) | (have_last ? OVW_F_LINK_B : 0u) | (r.present ? OVW_F_PRESENT : 0u);
        pend = ovw_pack(r.rec, (uint32_t)tn, flags, d_in, 0, r.lo);
        if (created && !ovw_post(ow.cur, r.rec, OVW_CREATION, ow.time, 1)) { wc.fail = 1; return; }
        if (d_in > ((r.lo >> (8 * ext_bwd)) & 0xFF) && !ovw_post(ow.cur, r.rec, (uint32_t)ext_bwd, ow.time, d_in)) { wc.fail = 1; return; }
        if (created) ovw_mark_created(wc, v, key);
        // the skip: the junction's forward distance as the sequential run shows it to this piece
        uint32_t d = (uint32_t)((r.lo >> (8 * ext_fwd)) & 0xFF);
        if (ow.have_prev) d = max(d, ovw_before(ow.prev, r.rec, (uint32_t)ext_fwd, ow.time));
        if (revisit) d = max(d, ovw_own(ow, r.rec, (uint32_t)ext_fwd, pend, true));
        have_last = true;
        last_lo = r.lo;
        last_t = tn;
        last_ext_fwd = ext_fwd;
        if (d < 1) d = 1;
        t = tn + (int)d;
        wc.nb_processed += 1;
        wc.nb_skipped += (unsigned long long)(d - 1);
        if (ow.overflow) { wc.fail = 1; return; }
    }

    if (!have_last) {   // add_fake_junction (ReadScanner.cpp:92-104)
        wc.nb_no_juncs++;
        const int len = (int)nwin + k - 1;
        const int m = len / 2 - k / 2;
        const uint64_t key = pv_kmer(v, wc.pl.codes, p0 + m, k);
        const int real = pv_base(v, wc.pl.codes, p0 + m + k);
        const int tm = 2 * m + 1;
        OvwRec r;
        if (!ovw_record(wc, key, r)) { wc.fail = 1; return; }
        const bool created = !r.present && !(ow.have_prev && ovw_before(ow.prev, r.rec, OVW_CREATION, ow.time));
        const uint32_t d_b = (uint32_t)((tm - 2 * j) & 0xFF), d_f = (uint32_t)(((2 * (int)nwin - 1 - tm) - 2 * j) & 0xFF);
        if (created && !ovw_post(ow.cur, r.rec, OVW_CREATION, ow.time, 1)) { wc.fail = 1; return; }
        if (d_b > ((r.lo >> 32) & 0xFF) && !ovw_post(ow.cur, r.rec, 4u, ow.time, d_b)) { wc.fail = 1; return; }
        if (d_f > ((r.lo >> (8 * real)) & 0xFF) && !ovw_post(ow.cur, r.rec, (uint32_t)real, ow.time, d_f)) { wc.fail = 1; return; }
        uint4 e = ovw_pack(r.rec, (uint32_t)tm, (created ? OVW_F_CREATED : 0u) | ((uint32_t)real << 1) | OVW_F_FAKE | OVW_F_DF | (r.present ? OVW_F_PRESENT : 0u), d_b, d_f, r.lo);
        ovw_emit(ow, e);
    } else {            // ReadScanner.cpp:202-206: the last junction's distance to the end of the read
        const uint32_t d_f = (uint32_t)(((2 * (int)nwin - 1 - last_t) - 2 * j) & 0xFF);
        pend.y |= OVW_F_DF << 16;
        pend.z = (pend.z & ~0xFFu) | d_f;
        if (d_f > ((last_lo >> (8 * last_ext_fwd)) & 0xFF) && !ovw_post(ow.cur, pend.x, (uint32_t)last_ext_fwd, ow.time, d_f)) { wc.fail = 1; return; }
        ovw_emit(ow, pend);
    }
    if (ow.overflow) wc.fail = 1;
    if (ow.have_old)
        for (uint32_t i = ow.n_new; i < ow.n_old; i++) {
            const uint4 o = ow.log[i];
            if (ovw_event_sig(o)) ovw_mark(ow, o);
        }
}

// a window begins: nothing listed, nothing settled, clean filters
__global__ void __launch_bounds__(256) k_ovw_reset(uint32_t* state, uint32_t* filt, uint32_t filt_words3, uint32_t* longp, uint32_t n_pieces,
                                                   uint32_t* marks, uint32_t mark_words3) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    if (i < 8 + OVW_MAX_ROUNDS) state[i] = 0;
    for (uint32_t a = i; a < filt_words3; a += stride) filt[a] = 0;
    for (uint32_t a = i; a < n_pieces; a += stride) longp[a] = 0;
    for (uint32_t a = i; a < mark_words3; a += stride) marks[a] = 0xFFFFFFFFu;
}

// One round: every listed piece of a cluster without long pieces walks on the events of the round before -- or, where no earlier piece has
// changed the events at any of its registered k-mers, keeps its log and posts its events again.
__global__ void __launch_bounds__(64) k_ovw_round(Planes pl, FdParams fp, JTable jt, const uint32_t* __restrict__ root, const WinDesc* __restrict__ wdp,
                                                  KoTables kt, OvwTables ot, uint32_t round, const uint32_t* __restrict__ bloom, DevCounters* cnt) {
    if ((kt.state[1] & 1u) || kt.state[3] == 0 || ot.state[1]) return;
    if (round >= 1 && ot.state[8 + round - 1] == 0) return;            // settled (round 0 always counts as a change)
    for (uint32_t a = blockIdx.x * blockDim.x + threadIdx.x; a < ot.filt_words; a += gridDim.x * blockDim.x) ot.filt_next[a] = 0;
    for (uint32_t a = blockIdx.x * blockDim.x + threadIdx.x; a <= ot.mark_mask; a += gridDim.x * blockDim.x) ot.mark_next[a] = 0xFFFFFFFFu;
    const uint32_t n = ot.state[0] < ot.list_cap ? ot.state[0] : ot.list_cap;
    bool changed = false, failed = false;
    unsigned long long kept = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) ot.state[2] = round + 1;
    const WinDesc wd = *wdp;
    for (uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x) {
        const uint32_t li = ot.list[idx];
        if (ot.longp[root[li]] || kt.bad[root[li]]) continue;      // (a cluster with a piece of more than 512 windows stays with k_walk, all of it)
        const uint2 pc = pl.pieces[wd.first_piece + li];
        const uint64_t lk0 = fd_bits_at(pl.lk, pc.x) & chunk_mask(pc.y, 0), lk1 = pc.y > 64 ? fd_bits_at(pl.lk, pc.x + 64) & chunk_mask(pc.y, 1) : 0ULL;
        uint4* const log = ot.log + 2ULL * kt.piece_base[li];
        uint32_t* const res = ot.res + 8 * idx;
        if (round > 0) {
            // has any EARLIER piece posted other events at one of this piece's registered k-mers than in the round before that?
            bool walk_again = false;
            for (uint32_t c = 0; c < 2 && !walk_again; c++)
                for (uint64_t w = c ? lk1 : lk0; w; w &= w - 1) {
                    const uint32_t h = pl.kh[pc.x + 64 * c + (uint32_t)__builtin_ctzll(w)];
                    if (ot.mark_prev[(h * 0x9E3779B1u) >> 8 & ot.mark_mask] < li) { walk_again = true; break; }
                }
            if (!walk_again) {       // the log stands; its events go into this round's tables
                const uint32_t n_log = res[0];
                for (uint32_t i = 0; i < n_log; i++)
                    if (!ovw_repost(ot.cur, log[i], li)) failed = true;
                kept++;
                continue;
            }
        }
        WalkCtx wc;
        wc.pl = pl; wc.fp = fp; wc.jt = jt; wc.cnt = cnt; wc.bloom = bloom; wc.win_seq = 0;
        wc.nb_processed = wc.nb_skipped = wc.nb_jcheck = wc.nb_no_juncs = wc.n_created = wc.n_filled = 0;
        wc.created_now = false; wc.fail = 0; wc.dbg = 0;
        OvwWalk ow;
        ow.prev = ot.prev; ow.cur = ot.cur;
        ow.have_prev = round > 0;
        ow.log = log;
        ow.have_old = round > 0;
        ow.cap = 2 * (uint32_t)(__popcll(lk0) + __popcll(lk1));
        ow.n_old = round > 0 ? res[0] : 0;
        ow.n_new = 0;
        ow.time = li;
        ow.changed = false;
        ow.overflow = false;
        ow.kh = pl.kh;
        ow.p0 = pc.x;
        ow.mark = ot.mark_cur;
        ow.mark_mask = ot.mark_mask;
        ovw_seen_clear();
        ovw_walk_piece(wc, ow, pc.x, pc.y);
        if (ow.n_new != ow.n_old || round == 0) ow.changed = true;
        res[0] = ow.n_new;
        res[1] = (uint32_t)wc.nb_processed; res[2] = (uint32_t)wc.nb_skipped; res[3] = (uint32_t)wc.nb_jcheck; res[4] = (uint32_t)wc.nb_no_juncs;
        res[5] = (uint32_t)wc.n_filled; res[6] = wc.fail == 4 ? 1u : 0u; res[7] = 0;
        changed |= ow.changed;
        failed |= wc.fail == 1;
    }
    if (__ballot(changed) && fd_lane() == 0) atomicOr(&ot.state[8 + round], 1u);
    if (__ballot(failed) && fd_lane() == 0) atomicOr(&ot.state[1], 1u);
    for (int o = 32; o > 0; o >>= 1) kept += __shfl_down(kept, o, 64);
    if (kept && fd_lane() == 0) atomicAdd(&cnt->ovw_kept, kept);
}

// Junction::addCoverage / update / the link flags of one log entry, on a record other pieces update at the same time
__device__ __forceinline__ void ovw_apply(uint64_t* rec_addr, uint32_t tn, uint32_t flags, uint32_t d_bwd, uint32_t d_fwd) {
    const bool fwd = tn & 1;
    const uint32_t real = (flags & OVW_F_REAL) >> 1, ib = fwd ? 4u : real, iff = fwd ? real : 4u;
    unsigned long long* w0 = (unsigned long long*)rec_addr;
    unsigned long long old = __hip_atomic_load(w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
        unsigned long long nw = old;
        if (((nw >> (8 * ib)) & 0xFF) < d_bwd) nw = (nw & ~(0xFFULL << (8 * ib))) | ((unsigned long long)d_bwd << (8 * ib));
        if ((flags & OVW_F_DF) && ((nw >> (8 * iff)) & 0xFF) < d_fwd) nw = (nw & ~(0xFFULL << (8 * iff))) | ((unsigned long long)d_fwd << (8 * iff));
        if (real < 3 && ((nw >> (8 * (5 + real))) & 0xFF) != 255) nw += 1ULL << (8 * (5 + real));       // coverage saturates at 255 (Junction.cpp:59-67)
        if (nw == old) break;
        const unsigned long long was = atomicCAS(w0, old, nw);
        if (was == old) break;
        old = was;
    }
    unsigned long long* w1 = w0 + 1;
    unsigned long long add = 0;
    if (flags & OVW_F_LINK_B) add |= 1ULL << (8 + ib);
    if (flags & OVW_F_LINK_F) add |= 1ULL << (8 + iff);
    if (real == 3 || add) {
        old = __hip_atomic_load(w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (;;) {
            unsigned long long nw = old | add;
            if (real == 3 && (nw & 0xFF) != 255) nw += 1;
            if (nw == old) break;
            const unsigned long long was = atomicCAS(w1, old, nw);
            if (was == old) break;
            old = was;
        }
    }
}

// the settled logs are applied; the counters of the settled paths are added up
__global__ void __launch_bounds__(64) k_ovw_commit(Planes pl, FdParams fp, JTable jt, const uint32_t* __restrict__ root, const uint32_t* __restrict__ count,
                                                   const WinDesc* __restrict__ wdp, uint64_t piece_seq_base, KoTables kt, OvwTables ot, int rounds,
                                                   DevCounters* cnt, int count_followers, const unsigned long long* __restrict__ ev_keys, uint64_t ev_entries,
                                                   uint64_t epoch_first) {
    if ((kt.state[1] & 1u) || kt.state[3] == 0) return;
    {   // How full did the event tables get?  The two tables hold what the window's last two posting rounds left (every round posts all events
        // again); counted here, behind the rounds, and kept as a maximum over the scan: the host grows the tables before a window overflows them
        // (an overflow is exact -- the window goes to the key-ordered walk -- and slow)
        unsigned n0 = 0, n1 = 0;
        for (uint64_t a = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; a < ev_entries; a += (uint64_t)gridDim.x * blockDim.x) {
            n0 += (ev_keys[a] >> 56) >= epoch_first ? 1u : 0u;
            n1 += (ev_keys[2 * ev_entries + a] >> 56) >= epoch_first ? 1u : 0u;
        }
        for (int o = 32; o > 0; o >>= 1) { n0 += __shfl_down(n0, o, 64); n1 += __shfl_down(n1, o, 64); }
        if (fd_lane() == 0) {
            if (n0) atomicAdd(&ot.state[6], n0);
            if (n1) atomicAdd(&ot.state[7], n1);
            __threadfence();
            if (atomicAdd(&ot.state[5], 1u) + 1u == gridDim.x * (blockDim.x / 64)) {
                const unsigned a = atomicAdd(&ot.state[6], 0u), b = atomicAdd(&ot.state[7], 0u);
                atomicMax(&cnt->ovw_fill, (unsigned long long)(a > b ? a : b));
            }
        }
    }
    const int settled = ovw_settled_round(ot.state, rounds);
    unsigned long long v[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // processed, skipped, jcheck, no_juncs, created, filled, walked, rounds
    if (settled >= 0) {
        const uint4* log = ot.log;
        const uint32_t* res_all = ot.res;
        const uint32_t n = ot.state[0] < ot.list_cap ? ot.state[0] : ot.list_cap;
        if (blockIdx.x == 0 && threadIdx.x == 0 && n) v[7] = (unsigned long long)settled + 1;
        const WinDesc wd = *wdp;
        for (uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x) {
            const uint32_t li = ot.list[idx];
            const uint32_t r = root[li];
            if (!ot.longp[r] && !kt.bad[r]) {
                if (r == li) {
                    if (count_followers) atomicAdd(&cnt->followers, (unsigned long long)count[r]);
                    atomicMax(&cnt->max_cluster, (unsigned long long)count[r] + 1);
                }
                const uint2 pc = pl.pieces[wd.first_piece + li];
                const uint32_t* res = res_all + 8 * idx;
                const uint4* lg = log + 2ULL * kt.piece_base[li];
                const uint64_t seq = piece_seq_base + wd.first_piece + li;
                for (uint32_t i = 0; i < res[0]; i++) {
                    const uint4 e = lg[i];
                    const uint32_t tn = e.y & 0xFFFF, flags = (e.y >> 16) & 0xFF;
                    const uint64_t pos = pc.x + (tn >> 1);
                    if (flags & OVW_F_CREATED) {   // JunctionMap::createJunction: one creator per key in a settled state
                        const uint64_t slot = e.x >> 1;
                        jt.stamps[e.x] = (seq << STAMP_SHIFT) | ((flags & OVW_F_FAKE) ? STAMP_FAKE : (uint64_t)tn);
                        atomicOr((unsigned long long*)&jt.keys[slot], 1ULL << (62 + (e.x & 1)));
                        const uint64_t hb = jt_filter_bit(jt, ld_agent(&jt.keys[slot]) & J_KEYMASK);
                        atomicOr(&jt.filter[hb >> 5], 1u << (hb & 31));
                        atomicOr(&pl.cr[pos >> 6], 1ULL << (pos & 63));
                        v[4]++;
                    }
                    if (pl.sF && !(flags & OVW_F_FAKE)) atomicOr(&((tn & 1) ? pl.sF : pl.sB)[pos >> 6], 1ULL << (pos & 63));   // result.push_back, :140
                    ovw_apply((uint64_t*)(jt.recs + (uint64_t)e.x * 16), tn, flags, e.y >> 24, e.z & 0xFF);
                }
                v[0] += res[1]; v[1] += res[2]; v[2] += res[3]; v[3] += res[4]; v[5] += res[5]; v[6] += 1;
                if (res[6]) atomicOr(&cnt->error_flags, 4ULL);   // a late junction test came out true on a settled path: the lazy scan is void (see fill_missing)
            }
        }
    }
    for (int c = 0; c < 8; c++)
        for (int o = 32; o > 0; o >>= 1) v[c] += __shfl_down(v[c], o, 64);
    if (fd_lane() == 0) {
        if (v[0]) atomicAdd(&cnt->nb_processed, v[0]);
        if (v[1]) atomicAdd(&cnt->nb_skipped, v[1]);
        if (v[2]) atomicAdd(&cnt->nb_jcheck, v[2]);
        if (v[3]) atomicAdd(&cnt->nb_no_juncs, v[3]);
        if (v[4]) atomicAdd(&cnt->n_junctions, v[4]);
        if (v[5]) atomicAdd(&cnt->flags_filled, v[5]);
        if (v[6]) { atomicAdd(&cnt->walk_parallel, v[6]); atomicAdd(&cnt->ovw[0], v[6]); }
        if (v[7]) { atomicAdd(&cnt->ovw[1], v[7]); atomicAdd(&cnt->ovw[2], 1ULL); }
        if (blockIdx.x == 0 && settled < 0 && kt.state[3]) atomicAdd(&cnt->ovw[3], 1ULL);   // a window left to the key-ordered walk
        // ... because its large clusters outgrew the optimistic walk's tables: a sign of far too large a window, like the key-ordered walk's own
        // overflow (the window-size controller reads ko_overflows; the pieces of these walks do not show as followers)
        if (blockIdx.x == 0 && settled < 0 && kt.state[3] && ot.state[1]) atomicAdd(&cnt->ko_overflows, 1ULL);
    }
}

// After the last window of a batch: the hashes of the keys its walk created, as a list the following batches register as their delta.
// (The planes themselves belong to a batch buffer that the pure stage recycles while later walks still run; the lists are the context's.)
// (run after every window over the words that window's pieces reach; the bits are cleared as they are taken, so a word that two
// consecutive windows share is collected twice without listing anything twice)
__global__ void __launch_bounds__(256) k_delta_collect(unsigned long long* __restrict__ cr, const uint32_t* __restrict__ kh, uint64_t w_first, uint64_t n_words,
                                                       uint32_t* __restrict__ list, unsigned long long* __restrict__ count, Planes pl,
                                                       const WinDesc* __restrict__ wdp, DevCounters* cnt, uint32_t win_seq) {
    __shared__ unsigned s_wave[4];
    __shared__ unsigned long long s_base;
    // The late junction tests of this window (fill_missing; nearly always none): does the k-mer of such a position occur anywhere else on
    // a piece of the window?  Then a piece outside the creating piece's cluster may have passed a junction it should have seen: void.
    __shared__ uint32_t s_late_h[FGPU_LATE_CAP];
    __shared__ uint64_t s_late_p[FGPU_LATE_CAP];
    __shared__ uint32_t s_late_n;
    if (threadIdx.x == 0) s_late_n = 0;
    // feedback for the window-span controller: the window's pieces are counted here, behind its walks, so that a snapshot of the counters
    // never holds a window's pieces without the followers among them (one writer per launch)
    if (blockIdx.x == 0 && threadIdx.x == 0) { cnt->walked_pieces += wdp->n; cnt->followers_seen = ld_agent((const uint64_t*)&cnt->followers); }
    __syncthreads();
    const unsigned long long n_noted = cnt->late_n[0] < FGPU_LATE_CAP ? cnt->late_n[0] : FGPU_LATE_CAP;
    if (n_noted) {
        if (threadIdx.x < n_noted && cnt->late[2 * threadIdx.x + 1] == win_seq) {
            const uint32_t at = atomicAdd(&s_late_n, 1u);
            s_late_p[at] = cnt->late[2 * threadIdx.x];
            s_late_h[at] = kh[cnt->late[2 * threadIdx.x]];
        }
        __syncthreads();
    }
    const uint32_t n_late = s_late_n;
    const int wave = (int)(threadIdx.x >> 6);
    for (uint64_t w0 = w_first + (uint64_t)blockIdx.x * 256; w0 < n_words; w0 += (uint64_t)gridDim.x * 256) {   // uniform trip count per block
        const uint64_t w = w0 + threadIdx.x;
        if (n_late && w < n_words) {
            const WinDesc wd = *wdp;
            uint64_t inside = pl.pm[w];
            while (inside) {
                const uint64_t pos = w * 64 + (uint64_t)__builtin_ctzll(inside);
                inside &= inside - 1;
                const uint32_t h = kh[pos];
                for (uint32_t e = 0; e < n_late; e++) {
                    uint32_t li;
                    uint2 pc;
                    if (h == s_late_h[e] && piece_in_window(pl, wd, pos, li, pc)) {
                        if (pos == s_late_p[e]) {
                            atomicAdd(&cnt->late_n[2], 1ULL);      // (the sweep has seen the noted position itself: fgpu_diag_late_flags)
                        } else {
                            atomicOr(&cnt->error_flags, 4ULL);
                            atomicAdd(&cnt->late_n[1], 1ULL);
                        }
                    }
                }
            }
        }
        unsigned long long m = w < n_words ? cr[w] : 0ULL;
        if (m) cr[w] = 0;
        unsigned mine = (unsigned)__popcll(m), incl = mine;
        for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(incl, o, 64); if (fd_lane() >= o) incl += t; }
        if (fd_lane() == 63) s_wave[wave] = incl;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
            s_base = total ? atomicAdd(count, (unsigned long long)total) : 0ULL;     // one reservation per block and round
        }
        __syncthreads();
        unsigned long long at = s_base + (incl - mine);
        for (int q = 0; q < wave; q++) at += s_wave[q];
        while (m) {
            const int b = __builtin_ctzll(m);
            m &= m - 1;
            list[at++] = kh[w * 64 + (uint64_t)b];
        }
        __syncthreads();
    }
}

// union-find and list entries of a window's pieces back to "every piece its own cluster": the arrays exist twice and consecutive
// windows alternate, so this runs on the side stream while the NEXT window is already being looked up and linked
__global__ void __launch_bounds__(256) k_walk_reset_uf(uint32_t* parent, uint32_t* count, uint32_t* head, uint32_t* par_fail, uint32_t n, uint4* wbits,
                                                       uint32_t wbits_vec) {
    for (uint32_t a = blockIdx.x * blockDim.x + threadIdx.x; a < n; a += gridDim.x * blockDim.x) {
        parent[a] = a;
        count[a] = 0;
        head[a] = U_INF;
        par_fail[a] = 0;
    }
    // ... and the window table's presence filter of this parity (a memset on the walk stream took 47 us per window when the
    // pure stage of the next batch ran beside it: 3.6 ms per step on the critical queue)
    for (uint32_t a = blockIdx.x * blockDim.x + threadIdx.x; a < wbits_vec; a += gridDim.x * blockDim.x) wbits[a] = make_uint4(0, 0, 0, 0);
}

__global__ void __launch_bounds__(256) k_fill_u64(uint64_t* p, uint64_t n, uint64_t v) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}
__global__ void __launch_bounds__(256) k_iota_u32(uint32_t* p, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = (uint32_t)i;
}

// FGPU_DEBUG_NEED_DROP=1 (tests): after the flags kernel, forget the evaluation of about half of the windows whose tests came out
// false -- need bit and NbJCheckKmer bits cleared -- so that the walk has to evaluate them itself wherever it scans them.  =2: see below.
// FGPU_DEBUG_WALK_STALL_US (tests): the walk stream held up for that long before every batch's walk, so that the pure stage gets as far ahead of
// the walk as the host lets it -- a bounded wait on the constant 100 MHz clock
__global__ void k_debug_stall(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

__global__ void __launch_bounds__(256) k_debug_need_drop(uint64_t n_words, uint64_t* __restrict__ ff, uint64_t* __restrict__ fb,
                                                         uint64_t* need, uint64_t* cf0, uint64_t* cf1, uint64_t* cb0, uint64_t* cb1, int mode,
                                                         const uint64_t* __restrict__ pm, const uint32_t* __restrict__ kh) {
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t drop = need[w] & ~(ff[w] | fb[w]) & fd_mix(w * 0x9E3779B97F4A7C15ULL + 12345);
        // mode 2: the evaluated positions of one k-mer in 16 (chosen by the k-mer's hash: all its occurrences alike) go whatever the answer
        // was, so that the walk meets tests that come out TRUE at k-mers nobody registered (fill_missing's late junction tests)
        if (mode == 2) {
            drop = 0;
            for (uint64_t m = need[w] & pm[w]; m; m &= m - 1) {
                const int b = __builtin_ctzll(m);
                if ((fd_mix((uint64_t)kh[w * 64 + (uint64_t)b] + 0x1234567ULL) & 15) == 0) drop |= 1ULL << b;
            }
        }
        need[w] &= ~drop;
        cf0[w] &= ~drop; cf1[w] &= ~drop; cb0[w] &= ~drop; cb1[w] &= ~drop;
        if (mode == 2) { ff[w] &= ~drop; fb[w] &= ~drop; }
    }
}

// ---- lazy flags: which positions can the walk ever stop skipping at? -------------------------------------------
// testForJunction is only ever evaluated at half-steps the walk SCANS (it skips `dist` half-steps after every junction).
// The pure stage therefore evaluates it only on a superset of the scanned positions, computed here per batch from whatever
// state the junction table is in at that moment (the previous batch's walk may still be running: every value read is a
// value the table had at some earlier time, and the argument below only needs the table to grow monotonically):
//   * a position in the map stays in the map; a scanned stretch ends at the first in-map position, now or later, so
//     junctions that appear later only shorten stretches that are already marked;
//   * a skip that lands exactly ON an in-map junction is final: every later piece reads the same distance, lands on the
//     same junction and re-links the same distance, so that distance is never raised;
//   * after any other skip (distance not yet converged) everything up to the end of the piece is marked.
// New junctions created inside a scanned stretch (flagged, spacer, fake) start with distance 0, i.e. the walk goes on
// scanning: they do not change which positions are visited.  The walk double-checks: before it scans a position whose need bit
// is clear it evaluates that position's tests itself (fill_missing); it never silently uses a flag that was not computed.
__global__ void __launch_bounds__(256) k_need_lookup(const uint64_t* __restrict__ codes, const uint64_t* __restrict__ pm, uint64_t n_words,
                                                     FdParams fp, JTable jt, uint64_t* __restrict__ nF, uint64_t* __restrict__ nB,
                                                     uint32_t* __restrict__ kh, uint64_t w_first) {
    const uint64_t total = n_words * 64;
    for (uint64_t p = w_first * 64 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (uint64_t)gridDim.x * blockDim.x) {
        bool inF = false, inB = false;
        if ((pm[p >> 6] >> (p & 63)) & 1ULL) {
            uint64_t km = fd_kmer_at(codes, p, fp.k);
            uint64_t rc = fd_revcomp(km, fp.k);
            uint64_t canon = km < rc ? km : rc;
            const uint32_t h32 = jt_h32(canon);
            kh[p] = h32;                     // the walk stage's kernels take the hash from here (see jt_h32)
            const uint64_t hb = jt_filter_bit_h(jt, h32);
            if ((jt.filter[hb >> 5] >> (hb & 31)) & 1u) {
                uint32_t present = jt_present_snapshot(jt, canon);
                inF = (present >> (km == canon ? 0 : 1)) & 1u;
                inB = (present >> (rc == canon ? 0 : 1)) & 1u;
            }
        }
        uint64_t mF = __ballot(inF), mB = __ballot(inB);
        if (fd_lane() == 0) { nF[p >> 6] = mF; nB[p >> 6] = mB; }
    }
}

// the two bits of a key in the filter of NEW keys (fgpu_scan_import_table; k_refresh_lookup probes it with the same rule): one word, two bits
__device__ __forceinline__ uint64_t delta_word(uint32_t h32, uint64_t bits_mask) {
    return ((((uint64_t)(h32 * 0x9E3779B1u) << 16) ^ (uint64_t)(h32 >> 7)) & bits_mask) >> 5;
}
__device__ __forceinline__ uint32_t delta_bits(uint32_t h32) { return (1u << (h32 & 31)) | (1u << ((h32 >> 22) & 31)); }

// The same planes made AGAIN for a batch whose hash plane exists (prepared batches of a read shard, right before their walk: the table has been
// replaced since the pure stage made them): the hash is read back instead of being extracted, mixed and written a second time (4 bytes per
// position of stores), the k-mer is only taken out where the filter says it may be in the map.  merge: the planes keep the bits they have
// (a filter of the keys that are NEW since the planes were made -- the table only grows).
__global__ void __launch_bounds__(256) k_refresh_lookup(const uint64_t* __restrict__ codes, const uint64_t* __restrict__ pm, uint64_t n_words,
                                                        FdParams fp, JTable jt, const uint32_t* __restrict__ filter, uint64_t filter_mask,
                                                        uint64_t* __restrict__ nF, uint64_t* __restrict__ nB, const uint32_t* __restrict__ kh, int merge,
                                                        uint64_t* __restrict__ cand) {
    // cand (merge only, may be null): the candidate plane of the sparse link pass gains every position that hits the filter of new keys
    const uint64_t total = n_words * 64;
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (uint64_t)gridDim.x * blockDim.x) {
        bool inF = false, inB = false, hitC = false;
        if ((pm[p >> 6] >> (p & 63)) & 1ULL) {
            const uint32_t h32 = kh[p];
            const uint64_t hb = ((((uint64_t)(h32 * 0x9E3779B1u) << 16) ^ (uint64_t)(h32 >> 7)) & filter_mask);
            const bool maybe = merge ? (filter[hb >> 5] & delta_bits(h32)) == delta_bits(h32) : (((filter[hb >> 5] >> (hb & 31)) & 1u) != 0);
            if (maybe) {
                hitC = true;
                const uint64_t km = fd_kmer_at(codes, p, fp.k);
                const uint64_t rc = fd_revcomp(km, fp.k);
                const uint64_t canon = km < rc ? km : rc;
                const uint32_t present = jt_present_snapshot(jt, canon);
                inF = (present >> (km == canon ? 0 : 1)) & 1u;
                inB = (present >> (rc == canon ? 0 : 1)) & 1u;
            }
        }
        const uint64_t mF = __ballot(inF), mB = __ballot(inB);
        if (cand) {
            const uint64_t mC = __ballot(hitC);
            if (fd_lane() == 0 && mC) cand[p >> 6] |= mC;
        }
        if (fd_lane() == 0) {
            if (merge) { if (mF) nF[p >> 6] |= mF; if (mB) nB[p >> 6] |= mB; }
            else { nF[p >> 6] = mF; nB[p >> 6] = mB; }
        }
    }
}

__global__ void k_dbg_diff(const uint64_t* a, const uint64_t* b, uint64_t n, unsigned long long* out, int slot) {
    unsigned long long d = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) d += a[i] != b[i];
    if (d) atomicAdd(&out[slot], d);
}
__device__ __forceinline__ void need_mark(unsigned long long* need, uint64_t a, uint64_t z) {   // positions [a, z)
    while (a < z) {
        uint64_t wi = a >> 6;
        uint64_t hi = (wi + 1) << 6;
        uint64_t upto = z < hi ? z : hi;
        int lo_b = (int)(a & 63), n_b = (int)(upto - a);
        unsigned long long m = (n_b == 64 ? ~0ULL : ((1ULL << n_b) - 1)) << lo_b;
        atomicOr(&need[wi], m);
        a = upto;
    }
}

__global__ void __launch_bounds__(256) k_need_prewalk(const uint64_t* __restrict__ codes, const uint2* __restrict__ pieces, uint64_t n_pieces,
                                                      FdParams fp, JTable jt, const uint64_t* __restrict__ nF, const uint64_t* __restrict__ nB,
                                                      unsigned long long* need, int tight) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pieces; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint2 pc = pieces[i];
        const uint64_t p0 = pc.x;
        const uint32_t nwin = pc.y;
        const int k = fp.k, j = fp.j;
        const int tmax = 2 * (int)nwin - 2 - 2 * j;
        int t = 2 * j + 1;
        while (t <= tmax) {
            // first in-map half-step >= t
            int t_ev = 0x7fffffff;
            const uint32_t q0 = (uint32_t)(t >> 1);
            for (uint32_t qc = q0; qc < nwin && 2 * (int)qc <= tmax; qc += 64) {
                uint64_t eF = fd_bits_at(nF, p0 + qc), eB = fd_bits_at(nB, p0 + qc);
                if (qc == q0 && (t & 1)) eB &= ~1ULL;
                const uint32_t rem = nwin - qc;
                if (rem < 64) { const uint64_t m = (1ULL << rem) - 1; eF &= m; eB &= m; }
                const int tb = eB ? 2 * (int)(qc + __builtin_ctzll(eB)) : 0x7fffffff;
                const int tf = eF ? 2 * (int)(qc + __builtin_ctzll(eF)) + 1 : 0x7fffffff;
                const int te = tb < tf ? tb : tf;
                if (te != 0x7fffffff) { t_ev = te; break; }
            }
            if (t_ev > tmax) {   // scans to the end of the piece
                need_mark(need, p0 + (uint64_t)(t >> 1), p0 + (uint64_t)(tmax >> 1) + 1);
                break;
            }
            // half-steps t .. t_ev - 1 are scanned (tested); the in-map junction's own half-step is not (find_next_junction looks the
            // key up before it tests, ReadScanner.cpp:63-66) -- marking its position as well cost 7 % more junction tests
            if (tight >= 1) { if (t < t_ev) need_mark(need, p0 + (uint64_t)(t >> 1), p0 + (uint64_t)((t_ev - 1) >> 1) + 1); }
            else need_mark(need, p0 + (uint64_t)(t >> 1), p0 + (uint64_t)(t_ev >> 1) + 1);
            // the junction's current skip distance
            const uint32_t q = (uint32_t)(t_ev >> 1);
            const bool fwd = t_ev & 1;
            const uint64_t km = fd_kmer_at(codes, p0 + q, k);
            const uint64_t rc = fd_revcomp(km, k);
            const uint64_t key = fwd ? km : rc;
            const uint64_t canon = km < rc ? km : rc;
            const int orient = key == canon ? 0 : 1;
            const int ext_fwd = fwd ? fd_base_at(codes, p0 + q + k) : 4;
            uint64_t s = fd_mix(canon) & jt.mask;
            int d = 1;
            for (uint64_t n = 0; n <= jt.mask; n++) {
                const uint64_t w = jt.keys[s];
                if (w == J_EMPTY) break;
                if ((w & J_KEYMASK) == canon) {
                    const uint8_t dv = jt.recs[(s * 2 + orient) * 16 + ext_fwd];
                    d = dv < 1 ? 1 : dv;
                    break;
                }
                s = (s + 1) & jt.mask;
            }
            const int land = t_ev + d;
            if (land > tmax) break;                       // jumps off the piece: nothing more is scanned
            const uint32_t lq = (uint32_t)(land >> 1);
            const uint64_t lbits = (land & 1) ? fd_bits_at(nF, p0 + lq) : fd_bits_at(nB, p0 + lq);
            // Lands between junctions (the distance still points at the end of an earlier read): the walk scans on from there.
            // Distances only grow, so the real landing is this one or later and the stretch marked from here covers it -- unless the
            // distance has meanwhile grown PAST the next junction of the snapshot, which the walk notices and repairs itself
            // (fill_missing).  tight < 2 is the older, conservative form: mark the rest of the piece (5 % more junction tests; the
            // repairs were as rare with either form: 0-4 windows per 10-20 M reads, scripts/lazy_flag_frequency.py).
            if (!(lbits & 1ULL) && tight < 2) {
                need_mark(need, p0 + lq, p0 + (uint64_t)(tmax >> 1) + 1);
                break;
            }
            t = land;                                     // lands on a junction: a final skip
        }
    }
}

// ---- export: compact the present records ------------------------------------------------------------------
struct ExportEntry {   // FGPU_TABLE_ENTRY_BYTES = 32
    uint64_t key;      // oriented k-mer
    uint64_t stamp;
    uint8_t rec[16];
};

__global__ void __launch_bounds__(256) k_export(JTable jt, FdParams fp, ExportEntry* out, uint64_t* stamps_out, unsigned long long* n_out) {
    // Every block owns a contiguous range of slots and reserves its output space ONCE: a first pass over the key words counts the
    // records of the range, one atomic claims that many entries, a second pass writes them (ranks from ballots and wave totals).
    // One same-address atomic per 256 slots, as before, cost 5 ms at 2^27 slots -- they serialise at ~10 ns each -- and the export is
    // on the path of every hand-over between ranks; the key words read twice are 1 GiB of streaming there.
    __shared__ unsigned wave_total[4];
    __shared__ unsigned long long block_base;
    const uint64_t n_slots = jt.mask + 1;
    const uint64_t per_block = ((n_slots + gridDim.x - 1) / gridDim.x + 255) & ~255ULL;
    const uint64_t lo = (uint64_t)blockIdx.x * per_block;
    const uint64_t hi = lo + per_block < n_slots ? lo + per_block : n_slots;
    const int wave = (int)(threadIdx.x >> 6);
    unsigned long long mine = 0;
    for (uint64_t s = lo + threadIdx.x; s < hi; s += 256) {
        const uint64_t w = jt.keys[s];
        if (w != J_EMPTY) mine += ((w >> 62) & 1ULL) + ((w >> 63) & 1ULL);
    }
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o, 64);
    if (fd_lane() == 0) wave_total[wave] = (unsigned)mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long total = (unsigned long long)wave_total[0] + wave_total[1] + wave_total[2] + wave_total[3];
        block_base = total ? atomicAdd(n_out, total) : 0ULL;
    }
    __syncthreads();
    unsigned long long base = block_base;
    for (uint64_t s0 = lo; s0 < hi; s0 += 256) {     // uniform trip count: hi - lo is a multiple of 256 except in the last block
        const uint64_t s = s0 + threadIdx.x;
        const uint64_t w = s < hi ? jt.keys[s] : J_EMPTY;
        const bool has0 = w != J_EMPTY && ((w >> 62) & 1ULL), has1 = w != J_EMPTY && ((w >> 63) & 1ULL);
        const uint64_t m0 = __ballot(has0), m1 = __ballot(has1);
        const uint64_t below = (1ULL << fd_lane()) - 1;
        unsigned rank = (unsigned)(__popcll(m0 & below) + __popcll(m1 & below));
        __syncthreads();                               // the totals of the previous chunk have been read by everybody
        if (fd_lane() == 0) wave_total[wave] = (unsigned)(__popcll(m0) + __popcll(m1));
        __syncthreads();
        for (int q = 0; q < wave; q++) rank += wave_total[q];
        unsigned long long idx = base + rank;
        base += (unsigned long long)wave_total[0] + wave_total[1] + wave_total[2] + wave_total[3];
        const uint64_t canon = w & J_KEYMASK;
        for (int o = 0; o < 2; o++) {
            if (!(o == 0 ? has0 : has1)) continue;
            ExportEntry e;
            e.key = o == 0 ? canon : fd_revcomp(canon, fp.k);
            e.stamp = jt.stamps[s * 2 + o];
            const uint64_t* r = (const uint64_t*)(jt.recs + (s * 2 + o) * 16);
            ((uint64_t*)e.rec)[0] = r[0];
            ((uint64_t*)e.rec)[1] = r[1];
            out[idx] = e;
            stamps_out[idx] = e.stamp;
            idx++;
        }
    }
}

__global__ void __launch_bounds__(256) k_gather_sorted(const ExportEntry* in, const uint32_t* order, uint64_t n, uint64_t* keys,
                                                       fgpu_junction* recs) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ExportEntry e = in[order[i]];
    keys[i] = e.key;
    fgpu_junction r;
    for (int c = 0; c < 4; c++) r.cov[c] = e.rec[5 + c];
    for (int c = 0; c < 5; c++) { r.dist[c] = e.rec[c]; r.linked[c] = (e.rec[9] >> c) & 1; }
    recs[i] = r;
}

// the keys a batch of this shard has just created join the filter of new keys: the batches behind it were prepared before ANY of this shard was
// walked, and their planes are only merged with what the filter names (the full refresh sees the table as it stands; this is its equal)
__global__ void __launch_bounds__(256) k_delta_filter_add(const uint32_t* __restrict__ list, const unsigned long long* __restrict__ count, uint32_t* dfilter,
                                                          uint64_t dmask) {
    const uint64_t n = *count;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t h32 = list[i];
        atomicOr(&dfilter[delta_word(h32, dmask)], delta_bits(h32));
    }
}

// what an import has to know about its entries beyond putting them into the table: the largest creation stamp (piece number) among them, and --
// when a preview is being replaced by a later state of the same table -- which keys are newer than the preview: counted, and put into the filter
__global__ void __launch_bounds__(256) k_import_probe(const ExportEntry* in, uint64_t n, FdParams fp, uint64_t after_seq, uint32_t* dfilter,
                                                      uint64_t dmask, unsigned long long* out) {
    // out[2], out[3]: a digest (XOR and sum of the mixed keys) of the entries that are NOT newer -- of a preview: of all its keys.  A table that is
    // a later state of a preview holds the preview's keys with their stamps, so its not-newer entries give the preview's digest (the
    // count alone could agree by coincidence for a table that is NOT such a state, e.g. after the sender replayed its scan)
    unsigned long long mx = 0, newer = 0, dx = 0, dsum = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const ExportEntry e = in[i];
        const unsigned long long seq = e.stamp >> STAMP_SHIFT;
        mx = seq > mx ? seq : mx;
        if (seq <= after_seq) {
            const unsigned long long m = fd_mix(e.key ^ 0x9E3779B97F4A7C15ULL);
            dx ^= m;
            dsum += m;
        }
        if (seq > after_seq) {
            newer++;
            if (dfilter) {
                const uint64_t rc = fd_revcomp(e.key, fp.k);
                const uint32_t h32 = jt_h32(e.key < rc ? e.key : rc);
                atomicOr(&dfilter[delta_word(h32, dmask)], delta_bits(h32));
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long t = __shfl_down(mx, o, 64);
        mx = t > mx ? t : mx;
        newer += __shfl_down(newer, o, 64);
        dx ^= __shfl_down(dx, o, 64);
        dsum += __shfl_down(dsum, o, 64);
    }
    if (fd_lane() == 0) {
        if (mx) atomicMax(&out[0], mx);
        if (newer) atomicAdd(&out[1], newer);
        if (dx) atomicXor(&out[2], dx);
        if (dsum) atomicAdd(&out[3], dsum);
    }
}

__global__ void __launch_bounds__(256) k_import(JTable jt, FdParams fp, const ExportEntry* in, uint64_t n, DevCounters* cnt) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ExportEntry e = in[i];
    uint64_t rc = fd_revcomp(e.key, fp.k);
    uint64_t canon = e.key < rc ? e.key : rc;
    int orient = e.key == canon ? 0 : 1;
    uint64_t slot;
    uint32_t present;
    const uint64_t home = fd_mix(canon) & jt.mask;
    if (!jt_find_or_claim(jt, canon, home, ld_agent(&jt.keys[home]), slot, present, cnt)) { atomicOr(&cnt->error_flags, 1ULL); return; }
    uint64_t* r = (uint64_t*)(jt.recs + (slot * 2 + orient) * 16);
    r[0] = ((const uint64_t*)e.rec)[0];
    r[1] = ((const uint64_t*)e.rec)[1];
    jt.stamps[slot * 2 + orient] = e.stamp;
    atomicOr((unsigned long long*)&jt.keys[slot], 1ULL << (62 + orient));
    const uint64_t hb = jt_filter_bit(jt, canon);
    atomicOr(&jt.filter[hb >> 5], 1u << (hb & 31));
}

static uint64_t jfilter_bits(fgpu_ctx* ctx) {
    static const int lg = getenv("FGPU_JFILTER_LOG2") ? atoi(getenv("FGPU_JFILTER_LOG2")) : 0;   // measurement aid
    const uint64_t full = ctx->jcap * 2;
    return lg >= 10 && (1ULL << lg) < full ? 1ULL << lg : full;
}
JTable make_jt(fgpu_ctx* ctx) { return JTable{ctx->jkeys, ctx->jrecs, ctx->jstamps, ctx->jcap - 1, ctx->jfilter, jfilter_bits(ctx) - 1}; }
WTable make_wt(fgpu_ctx* ctx, uint64_t epoch, int parity) {
    return WTable{ctx->wkeys, ctx->wbits + parity * ((1ULL << ctx->wbits_log2) / 32), ctx->wcap - 1, epoch << 56, (uint32_t)(32 - ctx->wbits_log2)};
}

}  // namespace

// fgpu_create touches one kernel of every translation unit from a helper thread: the runtime loads a unit's code object at the first use of
// one of its kernels (20-25 ms for the large units), which otherwise lands on the first batch of each pass
void fgpu_touch_scan_walk() {
    hipFuncAttributes attr;
    (void)hipFuncGetAttributes(&attr, (const void*)k_iota_u32);
}

// ---------------------------------------------------------------------------------------------------------------
int fgpu_scan_alloc(fgpu_ctx* ctx) {
    if (ctx->jkeys) return FGPU_OK;
    ctx->jcap = ctx->prm.junction_capacity;
    if (const char* e = getenv("FGPU_WBITS_LOG2")) ctx->wbits_log2 = std::min(28, std::max(16, atoi(e)));   // measurement aid
    // the key-ordered walk's tables (KoTables): 2^20 k-mers and 2^22 occurrences of large clusters per window, else the window is walked by cluster
    if (const char* e = getenv("FGPU_WALK_KO")) ctx->walk_ko = (uint32_t)std::max(0, atoi(e));
    ctx->walk_ko_always = getenv("FGPU_WALK_KO_ALWAYS") != nullptr || (ctx->prm.flags & FGPU_FLAG_KEY_ORDER_FROM_START) != 0;
    // a caller that expects repeats also gets the lower bar: with small batches a window holds only a few dozen pieces of a repeat's cluster
    if ((ctx->prm.flags & FGPU_FLAG_KEY_ORDER_FROM_START) && !getenv("FGPU_WALK_KO")) ctx->walk_ko = 32;
    // ... and the rule by weight: clusters whose pieces hold 128 lk positions and more between them (three pieces inside a repeat, thirty ordinary ones)
    ctx->walk_ko_weight = (ctx->prm.flags & FGPU_FLAG_KEY_ORDER_FROM_START) ? 128u : 0u;
    if (const char* e = getenv("FGPU_WALK_KO_WEIGHT")) ctx->walk_ko_weight = (uint32_t)std::max(0, atoi(e));
    if (const char* e = getenv("FGPU_OVW_EV_LOG2")) ctx->ovw_ev_log2 = std::min(27, std::max(8, atoi(e)));   // entries per event table to start with
    if (const char* e = getenv("FGPU_OVW_ROUNDS")) ctx->ovw_rounds = std::max(0, atoi(e));   // 0: the key-ordered walk takes every large cluster
    // (round 4: four times round 3's sizes -- the optimistic walk lets windows of repeat-rich data grow to 10^5 pieces of large clusters and more)
    ctx->ko_hk_cap = 1u << 22;
    ctx->ko_occ_cap = 1u << 24;
    if (const char* e = getenv("FGPU_WALK_HEAVY")) ctx->walk_heavy = (uint32_t)std::max(0, atoi(e));   // clusters of at least this many pieces are tried out of order; 0 = never
    // Scheduling windows span at most max_span positions (+ one piece length).  Worst case every position is a candidate with a distinct
    // k-mer, so the window table holds 4x that -- 32 bytes per position of the bound: 2 GiB at 2^26, 8 GiB at the 2^28 that filters of 2^32
    // bits and more ask for (fgpu_create).  The larger bounds are a matter of speed, never of results, so they give way to the memory there
    // is (several contexts on one device, or a smaller device, failed here where round 2 ran): the bound is first brought under
    // a quarter of the free device memory, and halved again -- everything allocated so far freed -- if an allocation fails all the same.
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
            while (ctx->max_span > FGPU_MAX_SPAN && 40ULL * ctx->max_span > free_b / 4) ctx->max_span >>= 1;
        (void)hipGetLastError();
    }
    for (;;) {
        while (ctx->max_span / (uint64_t)(ctx->fd.k + 1) + 2 >= W_OWNER_MASK) ctx->max_span >>= 1;   // piece indices of a window fit the table's owner field
        ctx->wcap = 4 * ctx->max_span;
        ctx->wmax = (uint32_t)(ctx->max_span / (uint64_t)(ctx->fd.k + 1) + 2);
        struct { void** p; uint64_t bytes; } want[] = {
            {(void**)&ctx->jkeys, ctx->jcap * 8}, {(void**)&ctx->jrecs, ctx->jcap * 32}, {(void**)&ctx->jstamps, ctx->jcap * 16}, {(void**)&ctx->jfilter, ctx->jcap * 2 / 8},
            {(void**)&ctx->wdesc, 64}, {(void**)&ctx->wkeys, ctx->wcap * 8},
            {(void**)&ctx->wbits, 2 * (1ULL << ctx->wbits_log2) / 8},                  // two filters: consecutive windows alternate
            {(void**)&ctx->uf_parent, 2ULL * ctx->wmax * 4},                            // two sets each: consecutive windows alternate (k_walk_reset_uf)
            {(void**)&ctx->cl_count, 2ULL * ctx->wmax * 4}, {(void**)&ctx->cl_offset, 2ULL * ctx->wmax * 4}, {(void**)&ctx->cl_fill, (uint64_t)ctx->wmax * 4},
            {(void**)&ctx->cl_fail, 2ULL * ctx->wmax * 4}, {(void**)&ctx->ko_hk, (uint64_t)ctx->ko_hk_cap * 4 * 3},
            {(void**)&ctx->ko_occ, (uint64_t)ctx->ko_occ_cap * (4 * 3 + 8)}, {(void**)&ctx->ko_piece, (uint64_t)ctx->wmax * 4 * 2 + 64},
            {(void**)&ctx->cl_members, (uint64_t)ctx->wmax * 4 * 2}, {(void**)&ctx->cl_roots, (uint64_t)ctx->wmax * 4 + 64}};
        hipError_t err = hipSuccess;
        uint64_t failed_bytes = 0;
        for (auto& w : want) {
            *w.p = nullptr;
            if ((err = hipMalloc(w.p, w.bytes)) != hipSuccess) { *w.p = nullptr; failed_bytes = w.bytes; break; }
        }
        if (err == hipSuccess) return FGPU_OK;
        (void)hipGetLastError();
        for (auto& w : want) { if (*w.p) hipFree(*w.p); *w.p = nullptr; }
        if (ctx->max_span <= FGPU_MAX_SPAN) {
            ctx->err = std::string("the scan's tables do not fit the device (hipMalloc of ") + std::to_string(failed_bytes) + " bytes: " + hipGetErrorString(err) + ")";
            return FGPU_ERR_NOMEM;
        }
        ctx->max_span >>= 1;                                                            // smaller windows: slower on thin coverage, same results
    }
}

int fgpu_scan_export_impl(fgpu_ctx* ctx, void* dev_entries, uint64_t cap_entries, uint64_t* d_stamps, uint64_t* n_entries);
int fgpu_scan_import_impl(fgpu_ctx* ctx, const void* dev_entries, uint64_t n);

// Rehash the junction table into new_cap slots.  Both streams are drained first; records, creation stamps and presence bits
// travel through the export format of the multi-GPU hand-over (k_export / k_import), so nothing about a junction changes.
int fgpu_scan_grow(fgpu_ctx* ctx, uint64_t new_cap) {
    if (new_cap <= ctx->jcap) return FGPU_OK;
    if (ctx->wstream) FGPU_HIP(fgpu_sync_stream(ctx, ctx->wstream));
    FGPU_HIP(fgpu_sync_stream(ctx, ctx->stream));
    FGPU_HIP(hipMemcpyAsync(&ctx->counters_host->n_junctions, &ctx->counters->n_junctions, 8, hipMemcpyDeviceToHost, ctx->stream));
    FGPU_HIP(hipMemcpyAsync(&ctx->counters_host->error_flags, &ctx->counters->error_flags, 8, hipMemcpyDeviceToHost, ctx->stream));
    FGPU_HIP(fgpu_sync_stream(ctx, ctx->stream));
    if (ctx->counters_host->error_flags & 1ULL) return FGPU_OK;   // already overflowed: the scan is void, the caller reports it
    const uint64_t n_max = ctx->counters_host->n_junctions + ctx->scan_imported;
    int rc;
    if ((rc = fgpu_ensure(ctx, &ctx->dl_entries, (n_max + 1) * sizeof(ExportEntry))) || (rc = fgpu_ensure(ctx, &ctx->dl_stamps, (n_max + 1) * 8))) return rc;
    uint64_t n = 0;
    if (n_max && (rc = fgpu_scan_export_impl(ctx, ctx->dl_entries.p, n_max, (uint64_t*)ctx->dl_stamps.p, &n))) return rc;
    if (n != n_max) { ctx->err = "junction count mismatch between counters and table (grow)"; return FGPU_ERR_STATE; }
    uint64_t* nk = nullptr; uint8_t* nr = nullptr; uint64_t* ns = nullptr; uint32_t* nf = nullptr;
    if (hipMalloc(&nk, new_cap * 8) != hipSuccess || hipMalloc(&nr, new_cap * 32) != hipSuccess || hipMalloc(&ns, new_cap * 16) != hipSuccess ||
        hipMalloc(&nf, new_cap * 2 / 8) != hipSuccess) {
        (void)hipGetLastError();
        hipFree(nk); hipFree(nr); hipFree(ns); hipFree(nf);
        ctx->err = "junction table: no memory to grow to " + std::to_string(new_cap) + " slots";
        return FGPU_ERR_NOMEM;
    }
    hipFree(ctx->jkeys); hipFree(ctx->jrecs); hipFree(ctx->jstamps); hipFree(ctx->jfilter);
    ctx->jkeys = nk; ctx->jrecs = nr; ctx->jstamps = ns; ctx->jfilter = nf;
    ctx->jcap = new_cap;
    FGPU_HIP(hipMemsetAsync(ctx->jkeys, 0xFF, ctx->jcap * 8, ctx->stream));
    FGPU_HIP(hipMemsetAsync(ctx->jrecs, 0, ctx->jcap * 32, ctx->stream));
    FGPU_HIP(hipMemsetAsync(ctx->jfilter, 0, ctx->jcap * 2 / 8, ctx->stream));
    if ((rc = fgpu_scan_import_impl(ctx, ctx->dl_entries.p, n))) return rc;
    FGPU_HIP(fgpu_sync_stream(ctx, ctx->stream));   // the walk stream starts on the new table
    ctx->scan_grown++;
    return FGPU_OK;
}

// A batch outgrew the table (error bit 1): the scan's attempt is void and is made again from the journal -- on a table with new_cap slots, EMPTY
// (nothing of the void attempt is kept).  Both streams idle.  The reference's unordered_map just grows (utils/JunctionMap.h:61).
int fgpu_scan_regrow_empty(fgpu_ctx* ctx, uint64_t new_cap) {
    if (new_cap <= ctx->jcap) return FGPU_OK;
    uint64_t* nk = nullptr; uint8_t* nr = nullptr; uint64_t* ns = nullptr; uint32_t* nf = nullptr;
    if (hipMalloc(&nk, new_cap * 8) != hipSuccess || hipMalloc(&nr, new_cap * 32) != hipSuccess || hipMalloc(&ns, new_cap * 16) != hipSuccess ||
        hipMalloc(&nf, new_cap * 2 / 8) != hipSuccess) {
        (void)hipGetLastError();
        hipFree(nk); hipFree(nr); hipFree(ns); hipFree(nf);
        ctx->err = "junction table full, and no memory to grow it to " + std::to_string(new_cap) + " slots";
        return FGPU_ERR_NOMEM;
    }
    hipFree(ctx->jkeys); hipFree(ctx->jrecs); hipFree(ctx->jstamps); hipFree(ctx->jfilter);
    ctx->jkeys = nk; ctx->jrecs = nr; ctx->jstamps = ns; ctx->jfilter = nf;
    ctx->jcap = new_cap;
    ctx->scan_grown++;
    return fgpu_scan_clear_table(ctx);
}

// empty junction table (a preview is being replaced by the real state); both streams idle
int fgpu_scan_clear_table(fgpu_ctx* ctx) {
    FGPU_HIP(hipMemsetAsync(ctx->jkeys, 0xFF, ctx->jcap * 8, ctx->stream));
    FGPU_HIP(hipMemsetAsync(ctx->jrecs, 0, ctx->jcap * 32, ctx->stream));
    FGPU_HIP(hipMemsetAsync(ctx->jfilter, 0, ctx->jcap * 2 / 8, ctx->stream));
    return FGPU_OK;
}

// called where the host holds a fresh count of the records (batch boundaries): keep the table below a quarter full
int fgpu_scan_reserve(fgpu_ctx* ctx, uint64_t records) {
    if (!ctx->jkeys || records * 4 <= ctx->jcap) return FGPU_OK;
    uint64_t want = ctx->jcap;
    while (want < records * 16 && want < (1ULL << 31)) want <<= 1;
    return fgpu_scan_grow(ctx, want);
}

int fgpu_scan_reset(fgpu_ctx* ctx) {
    FGPU_HIP(hipMemsetAsync(ctx->jkeys, 0xFF, ctx->jcap * 8, ctx->stream));
    FGPU_HIP(hipMemsetAsync(ctx->jrecs, 0, ctx->jcap * 32, ctx->stream));
    FGPU_HIP(hipMemsetAsync(ctx->jfilter, 0, ctx->jcap * 2 / 8, ctx->stream));
    if (!ctx->wt_epoch) FGPU_HIP(hipMemsetAsync(ctx->wkeys, 0, ctx->wcap * 8, ctx->stream));   // a new table: epoch 0 = no window's entry
    FGPU_HIP(hipMemsetAsync(ctx->wbits, 0, 2 * (1ULL << ctx->wbits_log2) / 8, ctx->stream));        // (a scan that broke off may have left bits behind)
    FGPU_LAUNCH("iota", k_iota_u32, 64, 256, ctx->uf_parent, (uint64_t)ctx->wmax);
    FGPU_LAUNCH("iota", k_iota_u32, 64, 256, ctx->uf_parent + ctx->wmax, (uint64_t)ctx->wmax);
    FGPU_HIP(hipMemsetAsync(ctx->cl_count, 0, 2 * ctx->wmax * 4, ctx->stream));
    FGPU_HIP(hipMemsetAsync(ctx->cl_offset, 0xFF, 2 * ctx->wmax * 4, ctx->stream));
    FGPU_HIP(hipMemsetAsync(ctx->cl_fail, 0, 2 * ctx->wmax * 4, ctx->stream));
    FGPU_HIP(hipMemsetAsync(ctx->ko_hk, 0xFF, (size_t)ctx->ko_hk_cap * 4 * 2, ctx->stream));        // keys free, lists empty,
    FGPU_HIP(hipMemsetAsync(ctx->ko_hk + 2 * (size_t)ctx->ko_hk_cap, 0, (size_t)ctx->ko_hk_cap * 4, ctx->stream));   // turns at zero,
    FGPU_HIP(hipMemsetAsync(ctx->ko_piece, 0, (size_t)ctx->wmax * 4 * 2 + 64, ctx->stream));        // flags and per-piece arrays clean
    return FGPU_OK;
}

// Pure-stage helper (main stream): need plane of the current batch = superset of the positions the walk will scan.
int fgpu_stage_scan_need(fgpu_ctx* ctx) {
    BatchBufs& bb = *ctx->cur;
    const uint64_t wb = (bb.n_words + FGPU_PADW) * 8;
    const bool eager = (ctx->prm.flags & FGPU_FLAG_EAGER_FLAGS) || ctx->eager_runtime || ctx->eager_scan;   // evaluate testForJunction everywhere
    FGPU_HIP(hipMemsetAsync(bb.need.p, eager ? 0xFF : 0, wb, ctx->stream));
    if (!bb.n_pieces) return FGPU_OK;
    if (int rc = fgpu_ensure_b(ctx, &bb.kh, (bb.n_words + FGPU_PADW) * 64 * 4)) return rc;

    static const int need_tight = getenv("FGPU_NEED_TIGHT") ? atoi(getenv("FGPU_NEED_TIGHT")) : 2;   // measurement aid, see k_need_prewalk
    JTable jt = make_jt(ctx);
    // also with eager flags: this kernel writes the hash plane the walk stage's kernels work from
    FGPU_LAUNCH("need_lookup", k_need_lookup, fgpu_grid(bb.n_words * 64, 256), 256, (const uint64_t*)bb.codes.p, (const uint64_t*)bb.pm.p,
                bb.n_words, ctx->fd, jt, (uint64_t*)bb.nF.p, (uint64_t*)bb.nB.p, (uint32_t*)bb.kh.p, (uint64_t)0);
    if (eager) return FGPU_OK;
    FGPU_LAUNCH("need_prewalk", k_need_prewalk, fgpu_grid(bb.n_pieces, 256), 256, (const uint64_t*)bb.codes.p, (const uint2*)bb.pieces.p,
                bb.n_pieces, ctx->fd, jt, (const uint64_t*)bb.nF.p, (const uint64_t*)bb.nB.p, (unsigned long long*)bb.need.p, need_tight);
    return FGPU_OK;
}

// Read shards, FGPU_PREPARED_REFRESH=overlap (fgpu_scan_walk_prepared; measured in round 5, not the default): the snapshot planes of a batch that
// was prepared before its turn, made again on the MAIN stream against the table as it stands, so that batch i + 1's are made while batch i is
// walked.  The batch's walk waits for pure_done, recorded here.
int fgpu_scan_refresh_planes(fgpu_ctx* ctx, BatchBufs* b) {
    if (!b->n_pieces || !b->T) return FGPU_OK;
    JTable jt = make_jt(ctx);
    ctx->launch_stream = ctx->stream;
    FGPU_LAUNCH("need_lookup", k_refresh_lookup, fgpu_grid(b->n_words * 64, 256), 256, (const uint64_t*)b->codes.p, (const uint64_t*)b->pm.p, b->n_words,
                ctx->fd, jt, (const uint32_t*)jt.filter, jt.filter_mask, (uint64_t*)b->nF.p, (uint64_t*)b->nB.p, (const uint32_t*)b->kh.p, 0, (uint64_t*)nullptr);
    if (b->pure_done) FGPU_HIP(hipEventRecord(b->pure_done, ctx->stream));
    return FGPU_OK;
}

// Read shards (fgpu_scan_refresh_prepared, off the chain): the candidate plane of a prepared batch against the planes as they stand -- see
// k_walk_link_sparse.  Two passes: the hashes of every candidate position and of the pieces' own candidates into a filter, then every position
// against that filter.
int fgpu_scan_build_cand(fgpu_ctx* ctx, BatchBufs* b) {
    b->cand_gen = 0;
    if (!b->n_pieces || !b->T) return FGPU_OK;
    int rc;
    const uint64_t wb = (b->n_words + FGPU_PADW) * 8;
    if ((rc = fgpu_ensure_b(ctx, &b->cand, wb)) || (rc = fgpu_ensure(ctx, &ctx->cand_filter, (1ULL << CAND_FILTER_LOG2) / 8))) return rc;
    ctx->launch_stream = ctx->stream;
    FGPU_HIP(hipMemsetAsync(ctx->cand_filter.p, 0, (1ULL << CAND_FILTER_LOG2) / 8, ctx->stream));
    FGPU_HIP(hipMemsetAsync(b->cand.p, 0, wb, ctx->stream));
    const unsigned word_blocks = std::min(fgpu_blocks(b->n_words, 256), 2048u), piece_blocks = std::min(fgpu_blocks(b->n_pieces, 256), 2048u);
    FGPU_LAUNCH("cand_plane", k_cand_mark, word_blocks + piece_blocks, 256, (const uint64_t*)b->pm.p, (const uint64_t*)b->nF.p, (const uint64_t*)b->nB.p,
                (const uint64_t*)b->ff.p, (const uint64_t*)b->fb.p, (const uint32_t*)b->kh.p, b->n_words, (const uint2*)b->pieces.p, b->n_pieces, ctx->fd,
                (uint32_t*)ctx->cand_filter.p, word_blocks);
    FGPU_LAUNCH("cand_plane", k_cand_probe, fgpu_grid(b->n_words * 64, 256), 256, (const uint64_t*)b->pm.p, (const uint32_t*)b->kh.p, b->n_words,
                (const uint32_t*)ctx->cand_filter.p, (uint64_t*)b->cand.p);
    if (b->pure_done) FGPU_HIP(hipEventRecord(b->pure_done, ctx->stream));      // (the walk waits for this event: the plane is part of what it reads)
    b->cand_gen = ctx->hint_gen;
    return FGPU_OK;
}

int fgpu_stage_scan_debug_drop(fgpu_ctx* ctx) {
    static const int drop = getenv("FGPU_DEBUG_NEED_DROP") ? atoi(getenv("FGPU_DEBUG_NEED_DROP")) : 0;
    BatchBufs& bb = *ctx->cur;
    if (!drop || !bb.n_pieces) return FGPU_OK;
    // (mode 2 makes the walk meet junction tests that come out true, which may void a LAZY scan; the eager scan that follows has to stand)
    if (drop == 2 && ((ctx->prm.flags & FGPU_FLAG_EAGER_FLAGS) || ctx->eager_runtime || ctx->eager_scan)) return FGPU_OK;
    FGPU_LAUNCH("debug_need_drop", k_debug_need_drop, fgpu_grid(bb.n_words, 256), 256, bb.n_words, (uint64_t*)bb.ff.p, (uint64_t*)bb.fb.p,
                (uint64_t*)bb.need.p, (uint64_t*)bb.cf0.p, (uint64_t*)bb.cf1.p, (uint64_t*)bb.cb0.p, (uint64_t*)bb.cb1.p, drop,
                (const uint64_t*)bb.pm.p, (const uint32_t*)bb.kh.p);
    return FGPU_OK;
}

// Walk the pieces of the current batch, scheduling window after scheduling window (position ranges of
// ctx->window_span stream positions).  No host round trip: window extents are derived on the device.
int fgpu_stage_scan_walk(fgpu_ctx* ctx, uint64_t n_pieces) {
    if (!n_pieces) return FGPU_OK;
    BatchBufs& bb = *ctx->cur;
    Planes pl{(const uint64_t*)bb.codes.p, (const uint64_t*)bb.pm.p, (const uint64_t*)bb.ps.p, (const uint32_t*)bb.ps_prefix.p,
              (const uint64_t*)bb.ff.p, (const uint64_t*)bb.fb.p, (const uint64_t*)bb.cf0.p, (const uint64_t*)bb.cf1.p,
              (const uint64_t*)bb.cb0.p, (const uint64_t*)bb.cb1.p, (uint64_t*)bb.nF.p, (uint64_t*)bb.nB.p, (const uint2*)bb.pieces.p,
              (uint64_t*)bb.lk.p, (const uint64_t*)bb.need.p, nullptr, nullptr, (const uint32_t*)bb.kh.p, nullptr};
    {   // in-map planes of the walk = the pure stage's snapshot planes nF / nB (see k_walk_register); creation plane of this batch
        const uint64_t wb = (bb.n_words + FGPU_PADW) * 8;
        if (int rc = fgpu_ensure_b(ctx, &bb.cr, wb)) return rc;
        pl.cr = (unsigned long long*)bb.cr.p;
    }
    if (ctx->record_stops) {
        const uint64_t wb = (bb.n_words + FGPU_PADW) * 8;
        int rc;
        if ((rc = fgpu_ensure_b(ctx, &bb.sF, wb)) || (rc = fgpu_ensure_b(ctx, &bb.sB, wb))) return rc;
        pl.sF = (unsigned long long*)bb.sF.p;
        pl.sB = (unsigned long long*)bb.sB.p;
    }
    JTable jt = make_jt(ctx);
    const uint64_t span = ctx->window_span;
    const uint64_t ext = bb.max_piece_span;          // a piece that starts inside the window may reach this far beyond it
    if (ctx->max_span + ext + 64 > ctx->wcap) {      // the per-position slot list of a window is sized for 4 x the largest span
        ctx->err = "a read of " + std::to_string(ext) + " bases is longer than the walk's window tables allow";
        return FGPU_ERR_CAPACITY;
    }
    if (ext >= (1ULL << (STAMP_SHIFT - 1)) - 64) {   // half-steps of a piece must fit the creation stamp's low bits (dump order)
        ctx->err = "a read of " + std::to_string(ext) + " bases is longer than the junction creation stamps allow (2^19 windows)";
        return FGPU_ERR_CAPACITY;
    }
    const uint64_t T = bb.T;
    const uint64_t seq_base = ctx->scan_piece_base;
    static const int dbg_walk = getenv("FGPU_DEBUG_WALK") ? atoi(getenv("FGPU_DEBUG_WALK")) : 0;
    static const bool serial_clean = getenv("FGPU_SERIAL_CLEAN") && getenv("FGPU_SERIAL_CLEAN")[0] == '1';   // measurement aid: union-find reset in line
    // the whole stage goes to the walk stream, behind the completion of this batch's pure stage
    static const bool no_overlap = getenv("FGPU_NO_OVERLAP") && getenv("FGPU_NO_OVERLAP")[0] == '1';   // measurement aid
    hipStream_t walk_stream = no_overlap ? ctx->stream : ctx->wstream;
    ctx->launch_stream = walk_stream;
    if (bb.pure_done) FGPU_HIP(hipStreamWaitEvent(walk_stream, bb.pure_done, 0));
    const int stall_us = ctx->dbg_stall_us;
    if (stall_us) FGPU_LAUNCH("debug_stall", k_debug_stall, 1, 1, (unsigned long long)stall_us * 100ULL);
    bool cand_ok = false;      // this batch's candidate plane covers every position its windows' link passes can find (see k_walk_link_sparse)
    if (ctx->refresh_snapshot) {
        // batches that were prepared before their turn (multi-GPU shards; scan_prepare / scan_walk_prepared) carry snapshot planes of a table
        // that has since been replaced or walked on by an unknown number of batches: made again here, behind the previous batch's walk
        // (the preview -- need plane and junction tests -- stays what it was: it is checked by the walk, not trusted)
        static const bool old_refresh = getenv("FGPU_REFRESH_OLD") != nullptr;      // measurement aid: the pure stage's kernel, as until round 5
        if (old_refresh)
            FGPU_LAUNCH("walk_lookup", k_need_lookup, fgpu_grid(bb.n_words * 64, 256), 256, (const uint64_t*)bb.codes.p, (const uint64_t*)bb.pm.p, bb.n_words,
                        ctx->fd, jt, (uint64_t*)bb.nF.p, (uint64_t*)bb.nB.p, (uint32_t*)bb.kh.p, (uint64_t)0);
        else if (ctx->delta_ready && bb.planes_gen == ctx->hint_gen) {
            // the planes speak of the newest preview and the table is a later state of it: only the keys created since are looked for and merged in
            // (and the candidate plane of the sparse link pass, made against the same preview, gains the positions that hit the filter: k_walk_link_sparse)
            cand_ok = !ctx->no_sparse_link && bb.cand.p && bb.cand_gen == ctx->hint_gen;      // (FGPU_NO_SPARSE_LINK: every window in full, as until round 6)
            if (ctx->delta_keys || ctx->refresh_delta)      // (nothing new in the table and nothing walked yet: the planes stand)
                FGPU_LAUNCH("walk_lookup", k_refresh_lookup, fgpu_grid(bb.n_words * 64, 256), 256, (const uint64_t*)bb.codes.p, (const uint64_t*)bb.pm.p, bb.n_words,
                            ctx->fd, jt, (const uint32_t*)ctx->delta_filter.p, ctx->delta_filter_bits - 1, (uint64_t*)bb.nF.p, (uint64_t*)bb.nB.p,
                            (const uint32_t*)bb.kh.p, 1, cand_ok ? (uint64_t*)bb.cand.p : (uint64_t*)nullptr);
            ctx->refresh_delta++;
            if (ctx->dbg_delta_check) {      // tests (FGPU_DEBUG_DELTA_CHECK=1): the merged planes against planes made again in full, word by word
                uint64_t *cF = nullptr, *cB = nullptr;
                unsigned long long* out = nullptr;
                FGPU_HIP(hipMalloc(&cF, bb.n_words * 8));
                FGPU_HIP(hipMalloc(&cB, bb.n_words * 8));
                FGPU_HIP(hipMalloc(&out, 32));
                FGPU_HIP(hipMemsetAsync(out, 0, 32, walk_stream));
                hipLaunchKernelGGL(k_refresh_lookup, dim3(fgpu_grid(bb.n_words * 64, 256)), dim3(256), 0, walk_stream, (const uint64_t*)bb.codes.p, (const uint64_t*)bb.pm.p,
                                   bb.n_words, ctx->fd, jt, (const uint32_t*)jt.filter, jt.filter_mask, cF, cB, (const uint32_t*)bb.kh.p, 0, (uint64_t*)nullptr);
                hipLaunchKernelGGL(k_dbg_diff, dim3(64), dim3(256), 0, walk_stream, (const uint64_t*)cF, (const uint64_t*)bb.nF.p, bb.n_words, out, 0);
                hipLaunchKernelGGL(k_dbg_diff, dim3(64), dim3(256), 0, walk_stream, (const uint64_t*)cB, (const uint64_t*)bb.nB.p, bb.n_words, out, 1);
                unsigned long long h[4] = {0, 0, 0, 0};
                FGPU_HIP(hipMemcpyAsync(h, out, 32, hipMemcpyDeviceToHost, walk_stream));
                FGPU_HIP(hipStreamSynchronize(walk_stream));
                ctx->refresh_mismatch += h[0] + h[1];
                if (h[0] + h[1]) fprintf(stderr, "[fgpu] merged in-map planes differ from planes made again: batch %llu, %llu + %llu words\n", (unsigned long long)bb.seq, h[0], h[1]);
                (void)hipFree(cF); (void)hipFree(cB); (void)hipFree(out);
            }
        } else {
            FGPU_LAUNCH("walk_lookup", k_refresh_lookup, fgpu_grid(bb.n_words * 64, 256), 256, (const uint64_t*)bb.codes.p, (const uint64_t*)bb.pm.p, bb.n_words,
                        ctx->fd, jt, (const uint32_t*)jt.filter, jt.filter_mask, (uint64_t*)bb.nF.p, (uint64_t*)bb.nB.p, (const uint32_t*)bb.kh.p, 0, (uint64_t*)nullptr);
            ctx->refresh_full++;
        }
    }
    if (ctx->record_stops) {
        const uint64_t wb = (bb.n_words + FGPU_PADW) * 8;
        FGPU_HIP(hipMemsetAsync(bb.sF.p, 0, wb, walk_stream));
        FGPU_HIP(hipMemsetAsync(bb.sB.p, 0, wb, walk_stream));
    }
    FGPU_HIP(hipMemsetAsync(bb.cr.p, 0, (bb.n_words + FGPU_PADW) * 8, walk_stream));
    {   // host-side estimate of the delta (see the per-window decision below): records counted when this batch's planes were made, and
        // how many the batches in the ring added
        const uint64_t now = ctx->counters_host->n_junctions;
        ctx->delta_hist[ctx->delta_next % FGPU_DELTA_RING] = now;
        const uint64_t back = ctx->delta_next >= (uint64_t)(FGPU_DELTA_RING - 1) ? ctx->delta_hist[(ctx->delta_next - (FGPU_DELTA_RING - 1)) % FGPU_DELTA_RING] : 0;
        ctx->delta_ring_keys = now > back ? now - back : 0;
        ctx->delta_batch_base = now;
    }
    // the list this batch's created keys go to: the oldest of the ring (its readers -- the windows of the batches in between -- are
    // behind us on this stream)
    DeltaList& mine_list = ctx->delta_ring[ctx->delta_next % FGPU_DELTA_RING];
    {
        int rc;
        if ((rc = fgpu_ensure_b(ctx, &mine_list.list, (bb.n_words + FGPU_PADW) * 64 * 4)) || (rc = fgpu_ensure(ctx, &mine_list.count, 64))) return rc;
        FGPU_HIP(hipMemsetAsync(mine_list.count.p, 0, 8, walk_stream));
    }
    // thousands of tiny launches: by default one event pair around the whole stage
    const int stage_tok = fgpu_prof_begin(ctx, "walk_stage");
    ctx->prof_suppress = !ctx->prof_walk_detail;
    uint64_t span_now = span;
    // A prepared batch with a candidate plane is one window when the controller's span would cut it into two or three: only a batch's FIRST
    // window is linked by the plane (the later ones register what the earlier ones created and are linked in full), and the walk itself is no
    // slower on the whole batch (hop of config 4 on 8 shards, 23 windows -> 14: k_walk_dyn 22.3 -> 22.6 ms, the link passes 12.7 -> 8.7, the hop
    // 69.1 -> 64.9 ms: profiles/r06_hop_sparse_link.txt).  The controller's own span is left as it is (it goes on voting from the counters; once it
    // asks for less than half a batch this rule no longer applies).  FGPU_NO_WHOLE_BATCH=1: off (measurement aid).
    static const bool no_whole = getenv("FGPU_NO_WHOLE_BATCH") != nullptr;
    const bool whole_batch = cand_ok && !no_whole && !ctx->prm.walk_window_span && ctx->calib_left <= 0 && span * 2 >= T && span < T && T <= ctx->max_span &&
                             bb.n_pieces <= ctx->wmax;
    for (uint64_t lo = 0, step = 0; lo < T; lo += step) {
        step = whole_batch ? T : span_now;
        const uint64_t hi = std::min<uint64_t>(T, lo + step);
        const uint64_t pos_end = std::min<uint64_t>(T, hi + ext);
        const int parity = (int)(ctx->scan_windows & 1);
        // grids sized for THIS window (a window of w positions holds at most w/(k+1)+2 piece starts): small windows -- high
        // coverage data -- must not pay for the launch of the thousands of empty blocks the largest window would need
        const uint64_t max_pieces = std::min<uint64_t>((hi - lo) / (uint64_t)(ctx->fd.k + 1) + 2, ctx->wmax);
        const unsigned walk_grid_w = fgpu_blocks(max_pieces, 64);
        static const unsigned cluster_cap = getenv("FGPU_CLUSTER_GRID") ? (unsigned)std::max(1, atoi(getenv("FGPU_CLUSTER_GRID"))) : 256u;
        const unsigned cluster_grid = std::min(cluster_cap, fgpu_blocks(max_pieces, 256));
        const unsigned piece_blocks_ko = fgpu_blocks(max_pieces, 256);
#ifdef FGPU_KO_ONE_XCD
        const unsigned ko_grid = 4096;
#else
        const unsigned ko_grid = 512;
#endif
        uint32_t* const uf_parent = ctx->uf_parent + parity * (uint64_t)ctx->wmax;     // this window's set of the union-find / list arrays
        uint32_t* const cl_count = ctx->cl_count + parity * (uint64_t)ctx->wmax;
        uint32_t* const cl_offset = ctx->cl_offset + parity * (uint64_t)ctx->wmax;
        uint32_t* const cl_fail = ctx->cl_fail + parity * (uint64_t)ctx->wmax;
        FGPU_HIP(hipStreamWaitEvent(walk_stream, ctx->ev_uf_reset[parity], 0));        // reset since the window before last used it
        // the window table of this window: entries carry the window's epoch, everything older counts as empty (wiped every 255 windows)
        if (++ctx->wt_epoch > 255) {
            FGPU_HIP(hipMemsetAsync(ctx->wkeys, 0, ctx->wcap * 8, walk_stream));
            ctx->wt_epoch = 1;
        }
        const WTable wt = make_wt(ctx, ctx->wt_epoch, parity);   // its presence filter was zeroed by the reset kernel of the window before last
        {
            DeltaSrc ds;
            memset(&ds, 0, sizeof(ds));
            // The delta of this window: keys created since the batch's snapshot planes were made -- by the batches walked since (their lists)
            // and by this batch's earlier windows (its own list so far).  Registering it costs one atomic pair per key and WINDOW; where the
            // windows are small and the delta is not (very high coverage: thousands of windows per batch, a million keys created by the
            // first of them) it is cheaper to make the snapshot planes of the window's own positions again -- what the look-up kernel of
            // round 1 did for every window -- and register no delta at all.  The host's estimate of the delta is an upper bound of what it
            // has seen counted; being wrong only picks the dearer of two exact ways.
            // (earlier batches: records counted between their walks' issue times; this batch: at most ~3 creations per piece walked so far)
            const uint64_t delta_est = ctx->delta_ring_keys + (uint64_t)(3.0 * (double)bb.n_pieces * (double)lo / (double)T);
            // (FGPU_REFRESH_DIV: the 4 below.  Swept in round 5 on configs 2, 5 and 4 at full size: 1, 2, 4 within noise of each other, 16 and more --
            // planes made again more readily -- slower everywhere: config 4's walk stage 1 202 -> 1 258 / 1 343 ms, config 2's step 119.7 -> 122.1 / 125.7 ms;
            // profiles/r05_refresh_div_sweep.txt)
            static const uint64_t refresh_div = getenv("FGPU_REFRESH_DIV") ? std::max(1, atoi(getenv("FGPU_REFRESH_DIV"))) : 4;
            const bool refresh_window = !ctx->refresh_snapshot && delta_est > (pos_end - lo) / refresh_div;
            if (refresh_window) {
                const uint64_t w_first = lo >> 6, w_last = std::min<uint64_t>(bb.n_words, (pos_end + 63) / 64);
                FGPU_LAUNCH("walk_lookup", k_need_lookup, fgpu_grid((w_last - w_first) * 64, 256), 256, (const uint64_t*)bb.codes.p, (const uint64_t*)bb.pm.p, w_last,
                            ctx->fd, jt, (uint64_t*)bb.nF.p, (uint64_t*)bb.nB.p, (uint32_t*)bb.kh.p, w_first);
            } else {
                if (!ctx->refresh_snapshot)      // (a batch whose planes were just made again needs no earlier batch's keys)
                    for (int r = 1; r < FGPU_DELTA_RING && (uint64_t)r <= ctx->delta_next; r++) {
                        const DeltaList& dl = ctx->delta_ring[(ctx->delta_next - r) % FGPU_DELTA_RING];
                        ds.list[r - 1] = (const uint32_t*)dl.list.p;
                        ds.count[r - 1] = (const unsigned long long*)dl.count.p;
                    }
                if (lo) {
                    ds.list[FGPU_DELTA_RING - 1] = (const uint32_t*)mine_list.list.p;
                    ds.count[FGPU_DELTA_RING - 1] = (const unsigned long long*)mine_list.count.p;
                }
            }
            const unsigned word_blocks = fgpu_blocks((pos_end - (lo & ~63ULL) + 63) / 64, 256);
            const unsigned piece_blocks = fgpu_blocks(max_pieces, 256);
            // the delta's share of the grid by its size: 64 blocks took a batch's worth of created keys (millions per window while a large map is
            // being built: config 4's first shards) 500 dependent atomics per thread, the kernel's tail.  One block per 1024 keys: walk stage of
            // config 4 1 216 -> 1 150-1 177 ms, config 5 298 -> 278-282 ms, the steps' wall time within noise (the pure stage beside it bounds pass 2);
            // config 2 unchanged (its delta never exceeds 64 blocks' worth).  profiles/r05_walk_grid_sweep.txt
            static const unsigned delta_per_block = getenv("FGPU_DELTA_PER_BLOCK") ? (unsigned)std::max(1, atoi(getenv("FGPU_DELTA_PER_BLOCK"))) : 1024u;
            const unsigned delta_blocks = refresh_window ? 64u : (unsigned)std::min<uint64_t>(4096, std::max<uint64_t>(64, delta_est / delta_per_block));
            FGPU_LAUNCH("walk_lookup", k_walk_register, word_blocks + piece_blocks + delta_blocks, 256, pl, ctx->fd, wt, uf_parent, lo, hi, pos_end, ctx->counters,
                        word_blocks, piece_blocks, ds);
        }
        uint32_t* const roots_state = ctx->cl_roots;               // [0] leaders listed, [1] handed out; the list follows (16 words in)
        uint32_t* const root_list = ctx->cl_roots + 16;
        if (cand_ok && lo == 0) {      // (lo > 0: the window registers what this batch's earlier windows created -- positions the plane does not know)
            const uint64_t link_words = ((pos_end + 63) >> 6) - (lo >> 6);
            FGPU_LAUNCH("walk_link", k_walk_link_sparse, std::min(fgpu_blocks(link_words, 256), 4096u), 256, pl, ctx->fd, wt, uf_parent, lo, hi, pos_end, roots_state,
                        (const uint64_t*)bb.cand.p, (uint64_t*)bb.lk.p);
            ctx->sparse_links++;
            if (ctx->dbg_delta_check) {      // tests: the full pass behind it (its unions are the same ones again), the two lk planes word by word
                uint64_t* copy = nullptr;
                unsigned long long* out = nullptr;
                FGPU_HIP(hipMalloc(&copy, link_words * 8));
                FGPU_HIP(hipMalloc(&out, 32));
                FGPU_HIP(hipMemsetAsync(out, 0, 32, walk_stream));
                FGPU_HIP(hipMemcpyAsync(copy, (const uint64_t*)bb.lk.p + (lo >> 6), link_words * 8, hipMemcpyDeviceToDevice, walk_stream));
                hipLaunchKernelGGL(k_walk_link, dim3(std::min(fgpu_blocks(pos_end - (lo & ~63ULL), 1024), 4096u)), dim3(256), 0, walk_stream, pl, ctx->fd, wt, uf_parent, lo, hi, pos_end,
                                   (uint32_t*)nullptr);
                hipLaunchKernelGGL(k_dbg_diff, dim3(64), dim3(256), 0, walk_stream, (const uint64_t*)copy, (const uint64_t*)bb.lk.p + (lo >> 6), link_words, out, 0);
                unsigned long long h[4] = {0, 0, 0, 0};
                FGPU_HIP(hipMemcpyAsync(h, out, 32, hipMemcpyDeviceToHost, walk_stream));
                FGPU_HIP(hipStreamSynchronize(walk_stream));
                ctx->refresh_mismatch += h[0];
                if (h[0]) fprintf(stderr, "[fgpu] sparse link pass: %llu lk words differ from the full pass's, batch %llu window at %llu\n", h[0], (unsigned long long)bb.seq, (unsigned long long)lo);
                (void)hipFree(copy); (void)hipFree(out);
            }
        } else {
            FGPU_LAUNCH("walk_link", k_walk_link, std::min(fgpu_blocks(pos_end - (lo & ~63ULL), 1024), 4096u), 256, pl, ctx->fd, wt, uf_parent, lo, hi, pos_end, roots_state);
            if (ctx->refresh_snapshot) ctx->full_links++;
        }
        // The key-ordered walk costs four small launches per window whether or not the window holds a large cluster, so it is switched on by
        // what the scan has shown so far: the largest cluster among the windows whose counters the host has seen (every batch's pure stage
        // brings them along).  Data without such clusters never pays; data with them walks its first batch by cluster.  Either way the
        // results are the same.  FGPU_WALK_KO_ALWAYS=1: from the first window (tests).
        const uint32_t ko_heavy = (dbg_walk || !ctx->walk_ko || !(ctx->walk_ko_always || ctx->counters_host->max_cluster >= ctx->walk_ko)) ? 0u : ctx->walk_ko;
        // ... and, for callers that expect repeats, the clusters whose pieces hold many lk positions between them (ko_cluster): the weights
        // are summed by k_walk_cluster into cl_fail, which the out-of-order walk (k_walk_par, the other way of taking clusters out of k_walk)
        // does not use then
        // The rule by weight has two bars: lk positions of a cluster in all (ko_total), and -- for clusters of "repeat pieces", pieces that hold
        // many of them -- positions per piece and a lower bar in all.  Ordinary pieces hold 2-5, some 8 and more; pieces inside a repeat at high
        // coverage 40 and more.  Once a scan HAS shown repeats (2 % of the pieces walked so far went to the large-cluster walks) the second bar
        // drops from (12 per piece, 48 in all) to (8, 16): measured on config 3's shape through the CLI pass 2 goes from 245-250 ms to 190 ms
        // (the windows grow from 62 to 22: what is left to k_walk no longer queues), while config 2 with the CLI's settings would pay 10 ms per
        // step for the lower bar (walk stage 52.6 -> 62.6 ms; scripts/ovw_thresholds.sh, gpurun_out/r04_ovw_thresholds2.txt).
        static const int ko_avg_env = getenv("FGPU_WALK_KO_AVG") ? std::min(255, std::max(0, atoi(getenv("FGPU_WALK_KO_AVG")))) : -1;
        static const int ko_avg_min_env = getenv("FGPU_WALK_KO_AVG_MIN") ? std::min(255, std::max(1, atoi(getenv("FGPU_WALK_KO_AVG_MIN")))) : -1;
        const bool repeats_seen = ctx->counters_host->walk_parallel * 50 > ctx->counters_host->walked_pieces ||
                                  (ctx->walk_ko_always && ctx->counters_host->walked_pieces < (1ULL << 17) && ctx->delta_next >= 2 && ctx->repeats_seen_before);   // (same lag: keep what the scan had found)
        if (repeats_seen) ctx->repeats_seen_before = true;
        const uint32_t ko_total = std::min<uint32_t>(ctx->walk_ko_weight, 0xFFFFu);
        const uint32_t ko_avg = ko_avg_env >= 0 ? (uint32_t)ko_avg_env : repeats_seen ? 8u : 12u;
        const uint32_t ko_total_rep = ko_avg_min_env > 0 ? (uint32_t)ko_avg_min_env : repeats_seen ? 16u : std::min<uint32_t>(255u, 3 * ko_total / 8);
        const uint32_t ko_heavy_w = (ko_heavy && ko_total) ? (ko_total | (ko_avg << 16) | (ko_total_rep << 24)) : 0u;
        // cl_count = followers per root, cl_offset = list heads, cl_fill = flat roots, cl_members = [next links | pool]
        FGPU_LAUNCH("walk_cluster", k_walk_cluster, cluster_grid, 256, (const uint32_t*)uf_parent, cl_count, cl_offset, ctx->cl_fill,
                    ctx->cl_members, pl, lo, hi, (WinDesc*)ctx->wdesc, ctx->counters, cl_fail, ko_heavy_w, root_list, roots_state);
        static const uint32_t ko_ticket = getenv("FGPU_KO_TICKET") ? (uint32_t)std::max(64, atoi(getenv("FGPU_KO_TICKET")) / 64 * 64) : 64u;
        const uint32_t heavy = (dbg_walk || ko_heavy) ? 0u : ctx->walk_heavy;   // (one way of taking large clusters out of k_walk at a time)
        KoTables kt;
        kt.hk_key = ctx->ko_hk;
        kt.hk_head = ctx->ko_hk + ctx->ko_hk_cap;
        kt.hk_turn = ctx->ko_hk + 2 * (size_t)ctx->ko_hk_cap;
        kt.hk_mask = ctx->ko_hk_cap - 1;
        kt.occ_id = (uint64_t*)ctx->ko_occ;
        kt.occ_entry = (uint32_t*)(kt.occ_id + ctx->ko_occ_cap);
        kt.occ_rank = kt.occ_entry + ctx->ko_occ_cap;
        kt.occ_next = kt.occ_rank + ctx->ko_occ_cap;
        kt.occ_cap = ctx->ko_occ_cap;
        kt.trace = nullptr;
#ifdef FGPU_KO_TRACE
        if (!ctx->ko_trace.p) {
            if (int trc = fgpu_ensure(ctx, &ctx->ko_trace, ((1ULL << 23) + 4096 * 1024) * 8)) return trc;
            FGPU_HIP(hipMemsetAsync(ctx->ko_trace.p, 0, 32, ctx->stream));
        }
        kt.trace = (unsigned long long*)ctx->ko_trace.p;
#endif
        kt.state = ctx->ko_piece;                 // 16 words in front of the per-piece arrays
        kt.piece_base = ctx->ko_piece + 16;
        kt.bad = kt.piece_base + ctx->wmax;
        // The optimistic walk (k_ovw_round) takes the large clusters first; the key-ordered walk runs behind it for what it leaves.  Its
        // rounds are launches whether or not the window holds such a cluster, so -- like the key-ordered walk itself -- it is switched on by
        // what the scan has shown: pieces of large clusters among the pieces walked so far (callers that expect repeats: from the first window).
        const int ovw_rounds = !ko_heavy ? 0 : std::min(ctx->ovw_rounds, OVW_MAX_ROUNDS);
        // the pieces it walks do not queue behind one another, so they do not count as followers for the window-size controller (the key-ordered
        // walk's did: 68 windows instead of 22 on config 3's shape); FGPU_OVW_FOLLOWERS=1 counts them (measurement aid)
        static const int ovw_followers = getenv("FGPU_OVW_FOLLOWERS") ? atoi(getenv("FGPU_OVW_FOLLOWERS")) : 0;
        // (a caller that expects repeats gets it for the scan's first two batches; after that the scan's own counters decide: on ordinary data the
        // few large clusters cost less with the key-ordered walk alone than with a dozen launches more per window -- config 2 with the CLI's
        // settings: walk stage 52.6 ms with the rounds always issued, scripts/ovw_config2.sh)
        // (the counters are the host's snapshot of the previous pure stage's end: when the walk lags two batches behind at that moment --
        // it happens, by the timing of the machine -- they do not show a single large cluster yet.  A caller that expects repeats therefore
        // keeps the rounds until the counters have SEEN enough pieces to say otherwise: round 4 met runs of config 3's shape in which the
        // third batch went without them and pass 2 took 310-360 instead of 160 ms)
        const bool ovw_seen_enough = ctx->counters_host->walked_pieces >= (1ULL << 17);
        static const bool dbg_ovw_gap = getenv("FGPU_DEBUG_OVW_GAP") != nullptr;   // measurement aid: the rounds withheld from a scan's third and fourth batch
        const bool ovw_on = ovw_rounds > 0 && !(dbg_ovw_gap && (ctx->delta_next == 2 || ctx->delta_next == 3)) &&
                            ((ctx->walk_ko_always && (ctx->delta_next < 2 || !ovw_seen_enough)) ||
                             ctx->counters_host->walk_parallel * 256 > ctx->counters_host->walked_pieces);
        OvwTables ot;
        memset(&ot, 0, sizeof(ot));
        {   // the event tables follow what the scan has shown (weak 9 of VERDICT r4): once a round of some window has filled a quarter of a table,
            // the next windows get tables eight times that round's entries -- before a window overflows them and is left to the key-ordered walk
            const uint64_t fill = ctx->counters_host->ovw_fill;
            int want = ctx->ovw_ev_log2;
            while (want < 27 && fill * 4 > (1ULL << want)) want++;
            if (want != ctx->ovw_ev_log2) {
                while (want < 27 && fill * 8 > (1ULL << want)) want++;
                if (ctx->ovw_ev.p) {     // rounds of the window before may still be reading the tables that are about to be replaced
                    FGPU_HIP(fgpu_sync_stream(ctx, walk_stream));
                    if (ctx->ostream) FGPU_HIP(fgpu_sync_stream(ctx, ctx->ostream));
                }
                ctx->ovw_ev_log2 = want;
            }
        }
        const int kEvLog2 = ctx->ovw_ev_log2, kFiltLog2 = kEvLog2 + 1;
        constexpr int kMarkLog2 = 22;
        const uint32_t filt_words = 1u << (kFiltLog2 - 5);
        const uint32_t list_cap = (uint32_t)std::min<uint64_t>(ctx->wmax, 1u << 20);
        uint32_t* ovw_filt = nullptr;
        if (ovw_on) {
            int rc;
            const bool fresh = !ctx->ovw_ev.p || ctx->ovw_ev_log2_alloc != kEvLog2;     // (grown since the last window: other table boundaries, wipe)
            ctx->ovw_ev_log2_alloc = kEvLog2;
            if ((rc = fgpu_ensure(ctx, &ctx->ovw_ev, 4ULL * 8 * (1ULL << kEvLog2))) || (rc = fgpu_ensure(ctx, &ctx->ovw_filt, 3ULL * 4 * filt_words)) ||
                (rc = fgpu_ensure(ctx, &ctx->ovw_log, 2ULL * ctx->ko_occ_cap * sizeof(uint4))) || (rc = fgpu_ensure(ctx, &ctx->ovw_res, 32ULL * list_cap)) ||
                (rc = fgpu_ensure(ctx, &ctx->ovw_marks, 3ULL * 4 * (1ULL << kMarkLog2))) ||
                (rc = fgpu_ensure(ctx, &ctx->ovw_list, 4ULL * list_cap)) || (rc = fgpu_ensure(ctx, &ctx->ovw_state, 4ULL * (8 + OVW_MAX_ROUNDS) + 64)) ||
                (rc = fgpu_ensure(ctx, &ctx->ovw_longp, 4ULL * ctx->wmax)))
                return rc;
            if (fresh || ctx->ovw_epoch + (uint32_t)ovw_rounds > 255) {   // event entries carry their round's epoch: a wiped table holds none
                FGPU_HIP(hipMemsetAsync(ctx->ovw_ev.p, 0, 4ULL * 8 * (1ULL << kEvLog2), walk_stream));
                ctx->ovw_epoch = 0;
            }
            ovw_filt = (uint32_t*)ctx->ovw_filt.p;
            ot.list = (uint32_t*)ctx->ovw_list.p;
            ot.list_cap = list_cap;
            ot.state = (uint32_t*)ctx->ovw_state.p;
            ot.longp = (uint32_t*)ctx->ovw_longp.p;
            ot.filt_words = filt_words;
        }
        if (ko_heavy) {
            FGPU_LAUNCH("walk_ko_prepare", k_ko_reset, 256, 256, kt, (uint32_t)ctx->wmax, (uint32_t)parity);
            if (ovw_on) FGPU_LAUNCH("walk_ko_prepare", k_ovw_reset, 64, 256, ot.state, ovw_filt, 3 * filt_words, ot.longp, (uint32_t)max_pieces,
                                    (uint32_t*)ctx->ovw_marks.p, 3u << kMarkLog2);
            FGPU_LAUNCH("walk_ko_prepare", k_ko_prepare, piece_blocks_ko, 256, pl, (const uint32_t*)ctx->cl_fill, (const uint32_t*)cl_count,
                        (const WinDesc*)ctx->wdesc, kt, ko_heavy, (uint32_t)parity, (const uint32_t*)cl_fail, ko_heavy_w, ot);
        }
        // The rounds run on a stream of their own BESIDE k_walk: the two walk disjoint clusters (a key belongs to the one cluster every piece that
        // holds it is in), so neither reads a record the other writes; slot claims, counters, planes and filters change by atomics on both sides.
        static const bool ovw_beside = !(getenv("FGPU_OVW_SERIAL") && getenv("FGPU_OVW_SERIAL")[0] == '1');   // measurement aid: behind k_walk, one stream
        hipStream_t ovw_stream = (ovw_beside && !no_overlap && ctx->ostream) ? ctx->ostream : walk_stream;
        if (ovw_on && ovw_stream != walk_stream) {
            FGPU_HIP(hipEventRecord(ctx->ev_listed, walk_stream));
            FGPU_HIP(hipStreamWaitEvent(ovw_stream, ctx->ev_listed, 0));
            ctx->launch_stream = ovw_stream;
        }
        if (ovw_on) {
            unsigned long long* const evb = (unsigned long long*)ctx->ovw_ev.p;
            uint32_t* const marks = (uint32_t*)ctx->ovw_marks.p;
            ot.log = (uint4*)ctx->ovw_log.p;
            ot.res = (uint32_t*)ctx->ovw_res.p;
            ot.mark_mask = (1u << kMarkLog2) - 1;
            const unsigned ovw_grid = (unsigned)std::min<uint64_t>(fgpu_blocks(std::min<uint64_t>(max_pieces, list_cap), 64), 2048);
            auto table = [&](int r, uint64_t epoch) {
                OvwEv ev;
                ev.key = evb + (uint64_t)(r & 1) * 2 * (1ULL << kEvLog2);
                ev.val = ev.key + (1ULL << kEvLog2);
                ev.bits = ovw_filt + (uint64_t)(r % 3) * filt_words;
                ev.mask = (1ULL << kEvLog2) - 1;
                ev.epoch = epoch << 56;
                ev.fshift = 32 - kFiltLog2;
                return ev;
            };
            const uint64_t epoch_first = (uint64_t)ctx->ovw_epoch + 1;
            for (int r = 0; r < ovw_rounds; r++) {
                ctx->ovw_epoch++;
                ot.cur = table(r, ctx->ovw_epoch);
                ot.prev = table(r + 1, ctx->ovw_epoch - 1);           // (the other table, the filter of round r - 1: (r + 2) % 3 == (r - 1) % 3)
                ot.prev.bits = ovw_filt + (uint64_t)((r + 2) % 3) * filt_words;
                ot.filt_next = ovw_filt + (uint64_t)((r + 1) % 3) * filt_words;
                ot.mark_cur = marks + ((uint64_t)(r % 3) << kMarkLog2);
                ot.mark_prev = marks + ((uint64_t)((r + 2) % 3) << kMarkLog2);
                ot.mark_next = marks + ((uint64_t)((r + 1) % 3) << kMarkLog2);
                FGPU_LAUNCH("walk_ovw", k_ovw_round, ovw_grid, 64, pl, ctx->fd, jt, (const uint32_t*)ctx->cl_fill, (const WinDesc*)ctx->wdesc, kt, ot, (uint32_t)r,
                            (const uint32_t*)ctx->bloo2, ctx->counters);
            }
            FGPU_LAUNCH("walk_ovw", k_ovw_commit, ovw_grid, 64, pl, ctx->fd, jt, (const uint32_t*)ctx->cl_fill, (const uint32_t*)cl_count, (const WinDesc*)ctx->wdesc,
                        seq_base, kt, ot, ovw_rounds, ctx->counters, ovw_followers, (const unsigned long long*)evb, (uint64_t)1 << kEvLog2, epoch_first);
        }
        if (ovw_on && ovw_stream != walk_stream) {
            FGPU_HIP(hipEventRecord(ctx->ev_settled, ovw_stream));
            ctx->launch_stream = walk_stream;
        }
        if (heavy)
            FGPU_LAUNCH("walk_probe", k_walk_par<WALK_PROBE>, walk_grid_w, 64, pl, ctx->fd, jt, (const uint32_t*)ctx->cl_fill, (const uint32_t*)cl_count, cl_fail,
                        (const WinDesc*)ctx->wdesc, (const uint32_t*)ctx->bloo2, ctx->counters, heavy);
        static const bool walk_static = getenv("FGPU_WALK_STATIC") && getenv("FGPU_WALK_STATIC")[0] == '1';   // measurement aid: lane i walks the cluster led by piece i
        static const unsigned walk_dyn_grid = getenv("FGPU_WALK_DYN_GRID") ? (unsigned)std::max(1, atoi(getenv("FGPU_WALK_DYN_GRID"))) : 1024u;   // (k_walk alone on config 2 / config 4's per-GPU shape, ms per step: static 19.3 / 22.9; 1024 waves 19.6 / 16.6; 2048 20.5 / 17.7; 4096 23.3 / 18.8)
        if (walk_static)
            FGPU_LAUNCH("walk", k_walk, walk_grid_w, 64, pl, ctx->fd, jt, (const uint32_t*)ctx->cl_fill, (const uint32_t*)cl_count,
                        (const uint32_t*)cl_offset, (const uint32_t*)ctx->cl_members, ctx->cl_members + ctx->wmax, (const WinDesc*)ctx->wdesc,
                        seq_base, (const uint32_t*)ctx->bloo2, ctx->counters, dbg_walk, (const uint32_t*)cl_fail, heavy, (const uint32_t*)kt.bad,
                        (const uint32_t*)kt.state, ko_heavy, (uint32_t)ctx->scan_windows, ko_heavy_w);
        else
            FGPU_LAUNCH("walk", k_walk_dyn, std::min(walk_grid_w, walk_dyn_grid), 64, pl, ctx->fd, jt, (const uint32_t*)root_list, roots_state, (const uint32_t*)cl_count,
                        (const uint32_t*)cl_offset, (const uint32_t*)ctx->cl_members, ctx->cl_members + ctx->wmax, (const WinDesc*)ctx->wdesc,
                        seq_base, (const uint32_t*)ctx->bloo2, ctx->counters, dbg_walk, (const uint32_t*)cl_fail, heavy, (const uint32_t*)kt.bad,
                        (const uint32_t*)kt.state, ko_heavy, (uint32_t)ctx->scan_windows, ko_heavy_w);
        if (ovw_on && ovw_stream != walk_stream) FGPU_HIP(hipStreamWaitEvent(walk_stream, ctx->ev_settled, 0));   // the settled logs are in the map
        if (ko_heavy)      // (ranks of the occurrences per k-mer: only the key-ordered walk needs them, and only for what the optimistic walk left)
            FGPU_LAUNCH("walk_ko_prepare", k_ko_rank, 256, 256, kt, (const uint32_t*)ot.state, ovw_rounds);
        if (ko_heavy)
            FGPU_LAUNCH("walk_ko", k_walk_ko, ko_grid, 64, pl, ctx->fd, jt, (const uint32_t*)ctx->cl_fill, (const uint32_t*)cl_count, (const WinDesc*)ctx->wdesc,
                        seq_base, (const uint32_t*)ctx->bloo2, ctx->counters, kt, ko_heavy, ko_ticket, (const uint32_t*)cl_fail, ko_heavy_w,
                        (const uint32_t*)ot.state, (const uint32_t*)ot.longp, ovw_rounds);
        if (heavy)
            FGPU_LAUNCH("walk_commit", k_walk_par<WALK_COMMIT>, walk_grid_w, 64, pl, ctx->fd, jt, (const uint32_t*)ctx->cl_fill, (const uint32_t*)cl_count, cl_fail,
                        (const WinDesc*)ctx->wdesc, (const uint32_t*)ctx->bloo2, ctx->counters, heavy);
        {   // the keys this window created join the batch's list (and leave the plane): the next window's delta
            const uint64_t w_first = lo >> 6, w_last = std::min<uint64_t>(bb.n_words, (pos_end + 63) / 64);
            FGPU_LAUNCH("walk_delta", k_delta_collect, (unsigned)std::min<uint64_t>(fgpu_blocks(w_last - w_first, 256), 256), 256, (unsigned long long*)bb.cr.p,
                        (const uint32_t*)bb.kh.p, w_first, w_last, (uint32_t*)mine_list.list.p, (unsigned long long*)mine_list.count.p, pl,
                        (const WinDesc*)ctx->wdesc, ctx->counters, (uint32_t)ctx->scan_windows);
        }
        // this window's set is reset on the side stream while the next window (the other set) is looked up and linked
        FGPU_HIP(hipEventRecord(ctx->ev_walked, walk_stream));
        FGPU_HIP(hipStreamWaitEvent(ctx->cstream, ctx->ev_walked, 0));
        ctx->launch_stream = ctx->cstream;
        FGPU_LAUNCH("walk_clean", k_walk_reset_uf, std::min(256u, fgpu_blocks(max_pieces, 256)), 256, uf_parent, cl_count, cl_offset, cl_fail, (uint32_t)max_pieces, (uint4*)wt.bits, (uint32_t)((1u << ctx->wbits_log2) / 128));
        ctx->launch_stream = walk_stream;
        FGPU_HIP(hipEventRecord(ctx->ev_uf_reset[parity], ctx->cstream));
        if (serial_clean) FGPU_HIP(hipStreamWaitEvent(walk_stream, ctx->ev_uf_reset[parity], 0));
        ctx->scan_windows++;
        // Calibration: at the start of a scan (and again whenever a batch came out with most of its pieces queueing) the host
        // waits for the window it has just issued and looks at the share of pieces that had to queue behind an earlier piece of
        // their cluster.  High coverage makes clusters percolate -- 250x reads of a small genome were 20x slower with the 4 M
        // position windows that suit 50x data -- and a whole batch is too long to walk with the wrong size.
        if (ctx->calib_left > 0 && !ctx->prm.walk_window_span) {
            FGPU_HIP(hipMemcpyAsync(ctx->fb_host, &ctx->counters->followers, 8, hipMemcpyDeviceToHost, walk_stream));
            FGPU_HIP(hipMemcpyAsync(ctx->fb_host + 1, &ctx->counters->walked_pieces, 8, hipMemcpyDeviceToHost, walk_stream));
            FGPU_HIP(hipMemcpyAsync(ctx->fb_host + 2, &ctx->counters->ko_overflows, 8, hipMemcpyDeviceToHost, walk_stream));
            FGPU_HIP(fgpu_sync_stream(ctx, walk_stream));
            const uint64_t f = ctx->fb_host[0] - ctx->calib_f, p = ctx->fb_host[1] - ctx->calib_p;
            if (ctx->fb_host[2] > ctx->calib_ovf && span_now > 4096) {
                // the window's large clusters outgrew the tables of the large-cluster walks (their pieces do not queue, so the share of
                // followers says nothing about them) and were walked by one thread each: far too large a window for this data -- a quarter,
                // and the next windows are looked at as well (round 4: read pairs inside repeats at 600x, a 36 MB file whose windows grew to
                // 2^23 positions within its three batches: pass 2 2.0 s, 0.8 s with windows of 2^20)
                ctx->calib_ovf = ctx->fb_host[2];
                ctx->calib_f = ctx->fb_host[0];
                ctx->calib_p = ctx->fb_host[1];
                span_now = std::max<uint64_t>(4096, span_now / 4);
                ctx->proven_span = std::min(ctx->proven_span, span_now);
                ctx->calib_left = std::max(ctx->calib_left, 4);
                ctx->span_ceiling = std::min(ctx->span_ceiling, span_now * 2);     // (this scan does not grow to that size again)
                static const bool dbg_span_o = getenv("FGPU_DEBUG_SPAN") != nullptr;
                if (dbg_span_o) fprintf(stderr, "[span] look: the large-cluster walks' tables overflowed -> span %llu, looks left %d\n", (unsigned long long)span_now, ctx->calib_left);
            } else
            if (p >= 64) {
                ctx->calib_f = ctx->fb_host[0];
                ctx->calib_p = ctx->fb_host[1];
                bool shrunk = false;
                if (f * 2 > p && span_now > 4096) { span_now /= 2; shrunk = true; }
                else if (f * 3 < p && span_now < std::min<uint64_t>(std::min<uint64_t>(ctx->max_span, FGPU_USUAL_SPAN), ctx->span_ceiling))
                    span_now = std::min<uint64_t>(span_now * 4, std::min<uint64_t>(std::min<uint64_t>(ctx->max_span, FGPU_USUAL_SPAN), ctx->span_ceiling));
                else if (f * 16 < p && span_now < std::min<uint64_t>(ctx->max_span, ctx->span_ceiling))
                    span_now = std::min<uint64_t>(span_now * 4, std::min<uint64_t>(ctx->max_span, ctx->span_ceiling));
                else ctx->calib_left = 1;          // settled
                ctx->calib_left--;
                // waiting for a window keeps the host from feeding the pure stage of the next batch: worth it while windows are small
                // (a fraction of a millisecond each, and the wrong size hurts most there); from 2^22 positions on the per-batch
                // controller (adapt_window, no waiting) takes over unless the last look said "too large"
                if (!shrunk && span_now >= (1ULL << 22)) ctx->calib_left = 0;
                static const bool dbg_span = getenv("FGPU_DEBUG_SPAN") != nullptr;
                if (dbg_span) fprintf(stderr, "[span] look: followers %llu of %llu pieces -> span %llu, looks left %d\n", (unsigned long long)f, (unsigned long long)p,
                                      (unsigned long long)span_now, ctx->calib_left);
            }
        }
    }
    if (ctx->delta_ready)      // (read shards, planes merged instead of made again: see k_delta_filter_add)
        FGPU_LAUNCH("walk_delta", k_delta_filter_add, 256, 256, (const uint32_t*)mine_list.list.p, (const unsigned long long*)mine_list.count.p,
                    (uint32_t*)ctx->delta_filter.p, ctx->delta_filter_bits - 1);
    ctx->delta_next++;
    ctx->window_span = span_now;
    ctx->prof_suppress = false;
    fgpu_prof_end(ctx, stage_tok);
    if (bb.walk_done) {
        FGPU_HIP(hipEventRecord(bb.walk_done, walk_stream));
        bb.walk_pending = true;
    }
    ctx->launch_stream = ctx->stream;
    ctx->scan_piece_base += n_pieces;
    return FGPU_OK;
}

// ---- scanInputRead's lists (FGPU_FLAG_RECORD_STOPS) ---------------------------------------------------------------
namespace {
__device__ __forceinline__ uint64_t piece_chunk(const unsigned long long* plane, uint64_t p0, uint32_t nwin, uint32_t c) {
    return fd_bits_at((const uint64_t*)plane, p0 + 64ULL * c) & chunk_mask(nwin, c);
}

// elements of the list of every piece: one per junction visit, or the single fake junction (ReadScanner.cpp:195-200)
__global__ void __launch_bounds__(256) k_stop_count(const uint2* __restrict__ pieces, uint64_t n_pieces, const unsigned long long* sF,
                                                    const unsigned long long* sB, uint32_t* __restrict__ count) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pieces) return;
    const uint2 pc = pieces[i];
    uint32_t n = 0;
    for (uint32_t c = 0; c * 64 < pc.y; c++) n += __popcll(piece_chunk(sF, pc.x, pc.y, c)) + __popcll(piece_chunk(sB, pc.x, pc.y, c));
    count[i] = n ? n : 1u;
}

__global__ void __launch_bounds__(256) k_stop_fill(const uint2* __restrict__ pieces, uint64_t n_pieces, const unsigned long long* sF,
                                                   const unsigned long long* sB, const uint32_t* __restrict__ offset,
                                                   const uint32_t* __restrict__ piece_read, const uint64_t* __restrict__ codes, FdParams fp,
                                                   fgpu_stop* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pieces) return;
    const uint2 pc = pieces[i];
    const uint64_t p0 = pc.x;
    const uint32_t nwin = pc.y, rd = piece_read[i];
    fgpu_stop* o = out + offset[i];
    uint32_t n = 0;
    for (uint32_t c = 0; c * 64 < nwin; c++) {
        const uint64_t wF = piece_chunk(sF, p0, nwin, c), wB = piece_chunk(sB, p0, nwin, c);
        uint64_t any = wF | wB;
        while (any) {
            const int b = __builtin_ctzll(any);
            any &= any - 1;
            const uint32_t q = c * 64 + b;
            if ((wB >> b) & 1ULL) {   // half-step 2q: facing BACKWARD, the extension is the reverse complement of the window before
                fgpu_stop e;
                e.ext = fd_revcomp(fd_kmer_at(codes, p0 + q - 1, fp.k), fp.k);
                e.read = rd;
                e.info = q | (n == 0 ? FGPU_STOP_FIRST : 0u);
                o[n++] = e;
            }
            if ((wF >> b) & 1ULL) {   // half-step 2q+1: facing FORWARD, the extension is the next window
                fgpu_stop e;
                e.ext = fd_kmer_at(codes, p0 + q + 1, fp.k);
                e.read = rd;
                e.info = q | FGPU_STOP_FORWARD | (n == 0 ? FGPU_STOP_FIRST : 0u);
                o[n++] = e;
            }
        }
    }
    if (n == 0) {   // add_fake_junction: middle k-mer facing FORWARD (ReadScanner.cpp:92-104)
        const uint32_t m = (nwin + (uint32_t)fp.k - 1) / 2 - (uint32_t)fp.k / 2;
        fgpu_stop e;
        e.ext = fd_kmer_at(codes, p0 + m + 1, fp.k);
        e.read = rd;
        e.info = m | FGPU_STOP_FORWARD | FGPU_STOP_FIRST | FGPU_STOP_FAKE;
        o[0] = e;
    }
}

// Bloom::addPair(JuncPair) (utils/Bloom.cpp:127-139): the two canonical k-mers, the smaller under seed 0 and the larger under seed 1
struct PairFilterDev {
    uint32_t* bits;
    uint64_t mask;
    int n_hash;
};
__device__ __forceinline__ void pf_add_pair(const PairFilterDev& pf, uint64_t k1, uint64_t k2, int k) {
    const uint64_t e1 = fd_canon(k1, k), e2 = fd_canon(k2, k);
    uint64_t h0 = fd_old_hash(e1 < e2 ? e1 : e2, FD_SEED0) & pf.mask;
    const uint64_t h1 = fd_old_hash(e1 < e2 ? e2 : e1, FD_SEED1) & pf.mask;
    for (int i = 0; i < pf.n_hash; i++) {
        atomicOr(&pf.bits[h0 >> 5], 1u << (h0 & 31));
        h0 = (h0 + h1) & pf.mask;
    }
}

// What scan_forward does with its result list when cleaning is on (src/ReadScanner.cpp:208-225), one thread per valid piece: a list of two
// is paired by facing (an outward-facing pair by first backward / last forward junction, two junctions facing the same way as they stand),
// a longer list pairs every element with the next but one.  Adding is order-free, so the device's order is as good as the file's.
__global__ void __launch_bounds__(256) k_short_pairs(const fgpu_stop* __restrict__ stops, const uint32_t* __restrict__ count,
                                                     const uint32_t* __restrict__ offset, uint64_t n_pieces, PairFilterDev pf, int k) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pieces) return;
    const uint32_t n = count[i];
    const fgpu_stop* s = stops + offset[i];
    if (n == 2) {
        bool have_first_back = false, have_last_fwd = false;
        uint32_t rev_pos = 0, for_pos = 0;
        uint64_t first_back = 0, last_fwd = 0;
        for (int t = 0; t < 2; t++) {
            const uint32_t info = s[t].info, pos = info & FGPU_STOP_POS_MASK;
            if (info & FGPU_STOP_FAKE) continue;
            if (!(info & FGPU_STOP_FORWARD)) {
                if (!have_first_back) { have_first_back = true; first_back = s[t].ext; rev_pos = pos; }
            } else {
                if (!have_last_fwd) { have_last_fwd = true; for_pos = pos; }
                last_fwd = s[t].ext;
            }
        }
        if (have_first_back && have_last_fwd && !(rev_pos > for_pos)) pf_add_pair(pf, first_back, last_fwd, k);
        if (have_first_back != have_last_fwd) pf_add_pair(pf, s[0].ext, s[1].ext, k);
    } else if (n > 2) {
        for (uint32_t t = 0; t + 2 < n; t++) pf_add_pair(pf, s[t].ext, s[t + 2].ext, k);
    }
}
}  // namespace

// Bring the stops of a walked batch to the host queue (waits for that batch's walk only).
int fgpu_scan_harvest(fgpu_ctx* ctx, BatchBufs* b) {
    for (size_t i = 0; i < ctx->to_harvest.size(); i++)
        if (ctx->to_harvest[i] == b) { ctx->to_harvest.erase(ctx->to_harvest.begin() + i); break; }
    if (!b->stops_pending) return FGPU_OK;
    // ONE wait of the host per harvested batch (round 5; there were three: for the walk's event, for the error flags, for the lists' total):
    // the main stream waits for the walk by itself, and what the host has to see -- the flags of the walk (a walk that went wrong must not hand
    // out lists), how many elements the lists hold -- comes back with one synchronisation.
    if (b->walk_pending) {
        FGPU_HIP(hipStreamWaitEvent(ctx->stream, b->walk_done, 0));
        b->walk_pending = false;
    }
    const bool check = ctx->journal_on && !ctx->in_replay;
    const uint64_t np_count = b->n_pieces;
    uint32_t* count_c = nullptr;
    uint32_t* offset_c = nullptr;
    if (np_count) {
        if (int rc0 = fgpu_ensure_b(ctx, &b->stop_off, (2 * np_count + 2) * 4)) return rc0;
        count_c = (uint32_t*)b->stop_off.p;
        offset_c = count_c + np_count + 1;
        hipLaunchKernelGGL(k_stop_count, dim3(fgpu_blocks(np_count, 256)), dim3(256), 0, ctx->stream, (const uint2*)b->pieces.p, np_count,
                           (const unsigned long long*)b->sF.p, (const unsigned long long*)b->sB.p, count_c);
        size_t tmp_bytes = 0;
        FGPU_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, count_c, offset_c, 0u, np_count, rocprim::plus<uint32_t>(), ctx->stream));
        DevBuf& tmp = ctx->probe_buf;
        if (int rc0 = fgpu_ensure_b(ctx, &tmp, tmp_bytes + 16)) return rc0;
        FGPU_HIP(rocprim::exclusive_scan(tmp.p, tmp_bytes, count_c, offset_c, 0u, np_count, rocprim::plus<uint32_t>(), ctx->stream));
        FGPU_HIP(hipMemcpyAsync(&ctx->fb_host[4], count_c + np_count - 1, 4, hipMemcpyDeviceToHost, ctx->stream));
        FGPU_HIP(hipMemcpyAsync(&ctx->fb_host[5], offset_c + np_count - 1, 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (check) FGPU_HIP(hipMemcpyAsync(&ctx->counters_host->error_flags, &ctx->counters->error_flags, 8, hipMemcpyDeviceToHost, ctx->stream));
    if (np_count || check) FGPU_HIP(fgpu_sync_stream(ctx, ctx->stream));
    if (check && (ctx->counters_host->error_flags & 4ULL)) { ctx->lazy_failed = true; ctx->to_harvest.insert(ctx->to_harvest.begin(), b); return FGPU_INTERNAL_REPLAY; }
    b->stops_pending = false;
    if (b->seq < ctx->stops_delivered) return FGPU_OK;   // scanned again by a replay: the caller has this batch's lists already
    // the lists come to the host unless their consumers on the device said otherwise (fgpu_scan_short_pairs' lists_to_host; a scan that only
    // feeds the long pair filter on the device, fgpu_scan_long_pairs, hands none out)
    const bool to_host = ctx->short_pf ? ctx->short_pf_lists_to_host : !ctx->lp.mode;
    if (to_host) {
        ctx->stop_queue.emplace_back();
        ctx->stop_queue.back().seq = b->seq;
    } else {
        ctx->stops_delivered = b->seq + 1;                // nobody takes them: the lists end in the device's pair filter
    }
    const uint64_t np = b->n_pieces;
    // the long pair filter takes every batch exactly once, whoever else reads the lists: with lists_to_host a replay harvests the batches the
    // caller has not taken yet again, and a second pass of the check-then-insert loop over them would count and insert twice
    const bool lp_due = b->seq >= ctx->lp_applied_seq;
    if (!np) {
        if (!lp_due) return FGPU_OK;
        const int rc0 = fgpu_long_pairs_batch(ctx, nullptr, 0, b->n_reads);   // (its reads still count as ends of pairs)
        if (!rc0) ctx->lp_applied_seq = b->seq + 1;
        return rc0;
    }
    int rc;
    uint32_t* count = count_c;
    uint32_t* offset = offset_c;
    const uint64_t total = (uint64_t)(uint32_t)ctx->fb_host[4] + (uint64_t)(uint32_t)ctx->fb_host[5];
    if ((rc = fgpu_ensure_b(ctx, &b->stop_out, total * sizeof(fgpu_stop)))) return rc;
    hipLaunchKernelGGL(k_stop_fill, dim3(fgpu_blocks(np, 256)), dim3(256), 0, ctx->stream, (const uint2*)b->pieces.p, np,
                       (const unsigned long long*)b->sF.p, (const unsigned long long*)b->sB.p, (const uint32_t*)offset,
                       (const uint32_t*)b->piece_read.p, (const uint64_t*)b->codes.p, ctx->fd, (fgpu_stop*)b->stop_out.p);
    if (ctx->short_pf)
        hipLaunchKernelGGL(k_short_pairs, dim3(fgpu_blocks(np, 256)), dim3(256), 0, ctx->stream, (const fgpu_stop*)b->stop_out.p, (const uint32_t*)count,
                           (const uint32_t*)offset, np, PairFilterDev{ctx->short_pf, ctx->short_pf_tai - 1, ctx->short_pf_hashes}, ctx->fd.k);
    // scanReads' paired-end loop over the same lists (pairs.hip): check-then-insert in file order, exact, on the device
    if (lp_due) {
        if ((rc = fgpu_long_pairs_batch(ctx, (const fgpu_stop*)b->stop_out.p, total, b->n_reads))) return rc;
        ctx->lp_applied_seq = b->seq + 1;
    }
    if (to_host) {
        StopBatch& sb = ctx->stop_queue.back();
        if (total) {
            size_t best = ctx->stop_pool.size();                       // the smallest free buffer that is large enough
            for (size_t i = 0; i < ctx->stop_pool.size(); i++)
                if (ctx->stop_pool[i].cap >= total && (best == ctx->stop_pool.size() || ctx->stop_pool[i].cap < ctx->stop_pool[best].cap)) best = i;
            if (best < ctx->stop_pool.size()) {
                sb.data = ctx->stop_pool[best].data;
                sb.cap = ctx->stop_pool[best].cap;
                ctx->stop_pool.erase(ctx->stop_pool.begin() + (ptrdiff_t)best);
            } else {
                const size_t cap = (size_t)(total + total / 4);
                FGPU_HIP(hipHostMalloc((void**)&sb.data, cap * sizeof(fgpu_stop)));
                sb.cap = cap;
            }
            FGPU_HIP(hipMemcpyAsync(sb.data, b->stop_out.p, total * sizeof(fgpu_stop), hipMemcpyDeviceToHost, ctx->stream));
            FGPU_HIP(fgpu_sync_stream(ctx, ctx->stream));
        }
        sb.n = total;
    }
    return FGPU_OK;
}

// ---- junction download / export / import -------------------------------------------------------------------------
int fgpu_scan_export_impl(fgpu_ctx* ctx, void* dev_entries, uint64_t cap_entries, uint64_t* d_stamps, uint64_t* n_entries) {
    unsigned long long* d_n = &ctx->counters->pad;
    FGPU_HIP(hipMemsetAsync(d_n, 0, 8, ctx->stream));
    (void)cap_entries;
    FGPU_LAUNCH("export", k_export, (unsigned)std::min<uint64_t>(fgpu_blocks(ctx->jcap, 256), 4096), 256, make_jt(ctx), ctx->fd, (ExportEntry*)dev_entries,
                d_stamps, d_n);
    FGPU_HIP(hipMemcpyAsync(&ctx->counters_host->pad, d_n, 8, hipMemcpyDeviceToHost, ctx->stream));
    FGPU_HIP(fgpu_sync_stream(ctx, ctx->stream));
    *n_entries = ctx->counters_host->pad;
    return FGPU_OK;
}

int fgpu_scan_import_impl(fgpu_ctx* ctx, const void* dev_entries, uint64_t n) {
    if (!n) return FGPU_OK;
    FGPU_LAUNCH("import", k_import, fgpu_blocks(n, 256), 256, make_jt(ctx), ctx->fd, (const ExportEntry*)dev_entries, n, ctx->counters);
    return FGPU_OK;
}

int fgpu_scan_import_probe(fgpu_ctx* ctx, const void* dev_entries, uint64_t n, uint64_t after_seq, uint32_t* dfilter, uint64_t dfilter_bits,
                           uint64_t* max_seq, uint64_t* n_newer, uint64_t digest[2]) {
    *max_seq = *n_newer = 0;
    digest[0] = digest[1] = 0;
    if (!n) return FGPU_OK;
    if (int rc = fgpu_ensure(ctx, &ctx->import_probe, 32)) return rc;
    FGPU_HIP(hipMemsetAsync(ctx->import_probe.p, 0, 32, ctx->stream));
    FGPU_LAUNCH("import", k_import_probe, (unsigned)std::min<uint64_t>(fgpu_blocks(n, 256), 4096), 256, (const ExportEntry*)dev_entries, n, ctx->fd, after_seq, dfilter,
                dfilter_bits ? dfilter_bits - 1 : 0, (unsigned long long*)ctx->import_probe.p);
    unsigned long long out[4] = {0, 0, 0, 0};
    FGPU_HIP(hipMemcpyAsync(out, ctx->import_probe.p, 32, hipMemcpyDeviceToHost, ctx->stream));
    FGPU_HIP(fgpu_sync_stream(ctx, ctx->stream));
    *max_seq = out[0];
    *n_newer = out[1];
    digest[0] = out[2];
    digest[1] = out[3];
    return FGPU_OK;
}

// ---- the reference's dump order on the device (host/junction_order.h: the closed form of what std::unordered_map does to its node list) -----
namespace {
// stretch j of the insertion sequence: position t holds node list[t] (t < c: the list as the last rehash found it) or node t itself (t >= c:
// inserted since); its bucket is key % B.  first[b] = the earliest position of bucket b.
__global__ void __launch_bounds__(256) k_dump_first(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ list, uint64_t c, uint64_t m, uint64_t B,
                                                    uint32_t* __restrict__ bucket, uint32_t* first) {
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < m; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t node = t < c ? list[t] : (uint32_t)t;
        const uint32_t b = (uint32_t)(keys[node] % B);
        bucket[t] = b;
        atomicMin(&first[b], (uint32_t)t);
    }
}
// sort key: (first time of the node's bucket, own time), both descending
__global__ void __launch_bounds__(256) k_dump_key(const uint32_t* __restrict__ list, uint64_t c, uint64_t m, const uint32_t* __restrict__ bucket,
                                                  const uint32_t* __restrict__ first, uint64_t* __restrict__ sortkey, uint32_t* __restrict__ node_out) {
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < m; t += (uint64_t)gridDim.x * blockDim.x) {
        node_out[t] = t < c ? list[t] : (uint32_t)t;
        sortkey[t] = ((uint64_t)(0xFFFFFFFFu - first[bucket[t]]) << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)t);
    }
}
}  // namespace

// order_host[i] = index (in creation order) of the i-th junction of the reference's dump; d_keys: the junction keys in creation order on the device
int fgpu_scan_dump_order_impl(fgpu_ctx* ctx, const uint64_t* d_keys, const uint64_t* counts, const uint64_t* buckets, uint64_t n_phases, uint64_t n,
                              uint32_t* order_host) {
    if (!n) return FGPU_OK;
    uint64_t max_b = 1;
    for (uint64_t j = 0; j < n_phases; j++) max_b = std::max(max_b, buckets[j]);
    DevBuf list_a, list_b, bucket, first, key_a, key_b, tmp;
    struct FreeAll { DevBuf* b[7]; ~FreeAll() { for (DevBuf* x : b) if (x->p) (void)hipFree(x->p); } } free_all{{&list_a, &list_b, &bucket, &first, &key_a, &key_b, &tmp}};
    auto alloc = [&](DevBuf& b, uint64_t bytes) -> int {
        if (hipMalloc(&b.p, bytes) != hipSuccess) { (void)hipGetLastError(); b.p = nullptr; ctx->err = "fgpu_scan_dump_order: out of device memory"; return FGPU_ERR_NOMEM; }
        b.bytes = bytes;
        return FGPU_OK;
    };
    int rc;
    if ((rc = alloc(list_a, n * 4)) || (rc = alloc(list_b, n * 4)) || (rc = alloc(bucket, n * 4)) || (rc = alloc(first, max_b * 4)) || (rc = alloc(key_a, n * 8)) ||
        (rc = alloc(key_b, n * 8)))
        return rc;
    size_t tmp_bytes = 0;
    FGPU_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, (uint64_t*)key_a.p, (uint64_t*)key_b.p, (uint32_t*)list_a.p, (uint32_t*)list_b.p, n, 0, 64, ctx->stream));
    if ((rc = alloc(tmp, tmp_bytes + 16))) return rc;
    uint32_t* list = (uint32_t*)list_a.p;      // the list as the last stretch left it
    uint32_t* other = (uint32_t*)list_b.p;
    bool any = false;
    for (uint64_t j = 0; j < n_phases; j++) {
        const uint64_t c = counts[j], m = j + 1 < n_phases ? counts[j + 1] : n, B = buckets[j];
        if (m == 0) continue;
        if (c > m || m > n || !B) { ctx->err = "fgpu_scan_dump_order: rehash counts must ascend to n, bucket counts must not be zero"; return FGPU_ERR_ARG; }
        FGPU_HIP(hipMemsetAsync(first.p, 0xFF, B * 4, ctx->stream));
        hipLaunchKernelGGL(k_dump_first, dim3(fgpu_grid(m, 256)), dim3(256), 0, ctx->stream, d_keys, (const uint32_t*)list, c, m, B, (uint32_t*)bucket.p, (uint32_t*)first.p);
        hipLaunchKernelGGL(k_dump_key, dim3(fgpu_grid(m, 256)), dim3(256), 0, ctx->stream, (const uint32_t*)list, c, m, (const uint32_t*)bucket.p, (const uint32_t*)first.p,
                           (uint64_t*)key_a.p, other);
        // (`other` holds the nodes in sequence order now; sorted into `list`)
        size_t tb = tmp_bytes;
        FGPU_HIP(rocprim::radix_sort_pairs(tmp.p, tb, (uint64_t*)key_a.p, (uint64_t*)key_b.p, other, list, m, 0, 64, ctx->stream));
        any = true;
    }
    if (!any) { ctx->err = "fgpu_scan_dump_order: empty schedule"; return FGPU_ERR_ARG; }
    FGPU_HIP(hipMemcpyAsync(order_host, list, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    FGPU_HIP(fgpu_sync_stream(ctx, ctx->stream));
    return FGPU_OK;
}

// creation-ordered download: export, radix-sort the stamps (rocPRIM; not a hot step), gather
int fgpu_scan_download_impl(fgpu_ctx* ctx, uint64_t* keys_host, fgpu_junction* recs_host, uint64_t cap, uint64_t* n_out) {
    uint64_t n_max = ctx->scan_stats.n_junctions;
    FGPU_HIP(hipMemcpyAsync(&ctx->counters_host->n_junctions, &ctx->counters->n_junctions, 8, hipMemcpyDeviceToHost, ctx->stream));
    FGPU_HIP(fgpu_sync_stream(ctx, ctx->stream));
    n_max = ctx->counters_host->n_junctions + ctx->scan_imported;
    *n_out = n_max;
    if (!n_max || !keys_host || !recs_host) return FGPU_OK;
    if (cap < n_max) { ctx->err = "junction buffer too small"; return FGPU_ERR_CAPACITY; }
    // scratch lives with the context: eight hipMalloc / hipFree pairs per call cost more than the sort they served
    int rc;
    if ((rc = fgpu_ensure(ctx, &ctx->dl_entries, n_max * sizeof(ExportEntry))) || (rc = fgpu_ensure(ctx, &ctx->dl_stamps, n_max * 8)) ||
        (rc = fgpu_ensure(ctx, &ctx->dl_stamps_sorted, n_max * 8)) || (rc = fgpu_ensure(ctx, &ctx->dl_idx, n_max * 4)) ||
        (rc = fgpu_ensure(ctx, &ctx->dl_idx_sorted, n_max * 4)) || (rc = fgpu_ensure(ctx, &ctx->dl_keys, n_max * 8)) ||
        (rc = fgpu_ensure(ctx, &ctx->dl_recs, n_max * sizeof(fgpu_junction))))
        return rc;
    ExportEntry* d_entries = (ExportEntry*)ctx->dl_entries.p;
    uint64_t *d_stamps = (uint64_t*)ctx->dl_stamps.p, *d_stamps_sorted = (uint64_t*)ctx->dl_stamps_sorted.p, *d_keys = (uint64_t*)ctx->dl_keys.p;
    uint32_t *d_idx = (uint32_t*)ctx->dl_idx.p, *d_idx_sorted = (uint32_t*)ctx->dl_idx_sorted.p;
    fgpu_junction* d_recs = (fgpu_junction*)ctx->dl_recs.p;
    uint64_t n = 0;
    if ((rc = fgpu_scan_export_impl(ctx, d_entries, n_max, d_stamps, &n))) return rc;
    if (n != n_max) { ctx->err = "junction count mismatch between counters and table"; return FGPU_ERR_STATE; }
    hipLaunchKernelGGL(k_iota_u32, dim3(256), dim3(256), 0, ctx->stream, d_idx, n);
    size_t tmp_bytes = 0;
    FGPU_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_stamps, d_stamps_sorted, d_idx, d_idx_sorted, n, 0, 64, ctx->stream));
    if ((rc = fgpu_ensure(ctx, &ctx->dl_tmp, tmp_bytes ? tmp_bytes : 16))) return rc;
    FGPU_HIP(rocprim::radix_sort_pairs(ctx->dl_tmp.p, tmp_bytes, d_stamps, d_stamps_sorted, d_idx, d_idx_sorted, n, 0, 64, ctx->stream));
    hipLaunchKernelGGL(k_gather_sorted, dim3(fgpu_blocks(n, 256)), dim3(256), 0, ctx->stream, (const ExportEntry*)d_entries,
                       (const uint32_t*)d_idx_sorted, n, d_keys, d_recs);
    // page-locked destinations (fgpu_host_alloc) receive this at link speed; pageable ones through the runtime's staging
    FGPU_HIP(hipMemcpyAsync(keys_host, d_keys, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    FGPU_HIP(hipMemcpyAsync(recs_host, d_recs, n * sizeof(fgpu_junction), hipMemcpyDeviceToHost, ctx->stream));
    FGPU_HIP(fgpu_sync_stream(ctx, ctx->stream));
    ctx->dl_keys_n = n;              // dl_keys holds the keys in creation order (fgpu_scan_dump_order works from them)
    return FGPU_OK;
}
