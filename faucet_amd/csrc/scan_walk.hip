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
//   A  k_walk_lookup   every position: look its canonical k-mer up in the junction table (snapshot before the
//                      window) -> in-map bit planes; register candidate k-mers in a small window table
//   B  k_walk_link     every position: probe the window table; a hit unions the piece with the candidate's owner
//   C  k_walk_cluster  flatten the union-find; list each cluster's members in ascending piece order
//   D  k_walk          one thread per cluster replays its pieces IN ORDER against the live table; clusters are
//                      disjoint in the keys they touch, so they run concurrently without changing any result
//   E  k_walk_clean    sparse reset of the window table
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

#include <rocprim/rocprim.hpp>

namespace {

constexpr uint64_t J_EMPTY = ~0ULL;
constexpr uint64_t J_KEYMASK = (1ULL << 62) - 1;
constexpr uint32_t U_INF = 0xFFFFFFFFu;

struct JTable {
    uint64_t* keys;
    uint8_t* recs;
    uint64_t* stamps;
    uint64_t mask;   // capacity - 1
};

struct WTable {
    uint64_t* keys;
    uint32_t* owner;
    uint32_t* slots;
    uint32_t* bits;      // 2^WBITS_LOG2-bit presence filter
    uint64_t mask;
};
constexpr int WBITS_LOG2 = 22;   // 4 Mbit = 512 KiB

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

// find or claim the slot of canon; returns false when the table is full
__device__ __forceinline__ bool jt_find_or_claim(const JTable& jt, uint64_t canon, uint64_t& slot, uint32_t& present, DevCounters* cnt) {
    uint64_t s = fd_mix(canon) & jt.mask;
    for (uint64_t n = 0; n <= jt.mask; n++) {
        uint64_t w = ld_agent(&jt.keys[s]);
        if (w == J_EMPTY) {
            unsigned long long old = atomicCAS((unsigned long long*)&jt.keys[s], (unsigned long long)J_EMPTY, (unsigned long long)canon);
            if (old == J_EMPTY) {
                atomicAdd(&cnt->table_slots_used, 1ULL);
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
// Returns the slot this call CLAIMED (the caller lists it for the sparse clean-up, one counter atomic per wave),
// or U_INF when the key was already there.
__device__ __forceinline__ uint32_t wt_register(const WTable& wt, uint32_t* parent, uint64_t canon, uint32_t piece, DevCounters* cnt) {
    uint64_t h = fd_mix(canon);
    uint64_t s = h & wt.mask;
    for (uint64_t n = 0; n <= wt.mask; n++) {
        unsigned long long old = atomicCAS((unsigned long long*)&wt.keys[s], (unsigned long long)J_EMPTY, (unsigned long long)canon);
        uint32_t claimed = U_INF;
        if (old == J_EMPTY) {
            claimed = (uint32_t)s;
            uint32_t b = (uint32_t)(h >> 40) & ((1u << WBITS_LOG2) - 1);
            atomicOr(&wt.bits[b >> 5], 1u << (b & 31));
            old = canon;
        }
        if (old == canon) {
            uint32_t prev = atomicMin(&wt.owner[s], piece);
            if (prev != U_INF && prev != piece) uf_union(parent, piece, prev);
            return claimed;
        }
        s = (s + 1) & wt.mask;
    }
    atomicOr(&cnt->error_flags, 2ULL);
    return U_INF;
}

__device__ __forceinline__ uint32_t wt_owner(const WTable& wt, uint64_t canon) {
    uint64_t h = fd_mix(canon);
    uint32_t b = (uint32_t)(h >> 40) & ((1u << WBITS_LOG2) - 1);
    if (!((wt.bits[b >> 5] >> (b & 31)) & 1u)) return U_INF;
    uint64_t s = h & wt.mask;
    for (uint64_t n = 0; n <= wt.mask; n++) {
        uint64_t w = wt.keys[s];
        if (w == J_EMPTY) return U_INF;
        if (w == canon) return wt.owner[s];
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

__global__ void __launch_bounds__(256) k_walk_lookup(Planes pl, FdParams fp, JTable jt, WTable wt, uint32_t* parent,
                                                     uint64_t lo, uint64_t hi, uint64_t pos_end, DevCounters* cnt, int parity) {
    const WinDesc wd = make_window(pl, lo, hi);
    uint64_t p = (wd.lo & ~63ULL) + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool inF = false, inB = false;
    uint32_t li;
    uint2 pc;
    uint32_t claimed = U_INF;
    if (p >= wd.lo && p < pos_end && ((pl.pm[p >> 6] >> (p & 63)) & 1ULL) && piece_in_window(pl, wd, p, li, pc)) {
        uint64_t km = fd_kmer_at(pl.codes, p, fp.k);
        uint64_t rc = fd_revcomp(km, fp.k);
        uint64_t canon = km < rc ? km : rc;
        uint32_t present = jt_present_snapshot(jt, canon);
        int oF = km == canon ? 0 : 1;   // orientation of the forward-facing key (the k-mer itself)
        int oB = rc == canon ? 0 : 1;   // orientation of the backward-facing key (its reverse complement)
        inF = (present >> oF) & 1u;
        inB = (present >> oB) & 1u;
        uint32_t q = (uint32_t)(p - pc.x);
        uint32_t len = pc.y + fp.k - 1;
        bool cand = present != 0;
        cand |= ((pl.ff[p >> 6] | pl.fb[p >> 6]) >> (p & 63)) & 1ULL;
        cand |= q == len / 2 - (uint32_t)fp.k / 2;                       // add_fake_junction's k-mer (ReadScanner.cpp:94)
        cand |= 2 * q + 1 >= (uint32_t)(2 * fp.max_spacer - 1);           // spacer rule can fire here (ReadScanner.cpp:72)
        if (cand) claimed = wt_register(wt, parent, canon, li, cnt);
    }
    uint64_t mF = __ballot(inF), mB = __ballot(inB);
    // list the slots this wave claimed: one counter atomic per wave
    uint64_t mC = __ballot(claimed != U_INF);
    unsigned long long base = 0;
    if (fd_lane() == 0) {
        pl.inF[p >> 6] = mF;
        pl.inB[p >> 6] = mB;
        if (mC) base = atomicAdd(parity ? &cnt->wt_used_b : &cnt->wt_used, (unsigned long long)__popcll(mC));
    }
    if (mC) {
        base = __shfl(base, 0, 64);
        if (claimed != U_INF) wt.slots[base + __popcll(mC & ((1ULL << fd_lane()) - 1))] = claimed;
    }
}

// ---- B: link every piece to the owners of the candidate k-mers that occur on it ---------------------
__global__ void __launch_bounds__(256) k_walk_link(Planes pl, FdParams fp, WTable wt, uint32_t* parent, uint64_t lo, uint64_t hi,
                                                   uint64_t pos_end) {
    const WinDesc wd = make_window(pl, lo, hi);
    uint64_t p = (wd.lo & ~63ULL) + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= wd.lo && p < pos_end && ((pl.pm[p >> 6] >> (p & 63)) & 1ULL)) {
        uint64_t canon = fd_canon(fd_kmer_at(pl.codes, p, fp.k), fp.k);
        uint32_t owner = wt_owner(wt, canon);
        if (owner != U_INF) {
            uint32_t li;
            uint2 pc;
            if (piece_in_window(pl, wd, p, li, pc) && owner != li) uf_union(parent, li, owner);
        }
    }
}

// ---- C: clusters -> ordered member lists (one block) --------------------------------------------------
constexpr int CL_BLOCK = 1024;
__global__ void __launch_bounds__(CL_BLOCK) k_walk_cluster(uint32_t* parent, uint32_t* count, uint32_t* offset, uint32_t* fill,
                                                           uint32_t* members, Planes pl, uint64_t lo, uint64_t hi, DevCounters* cnt) {
    __shared__ uint32_t sh[CL_BLOCK];
    __shared__ uint32_t carry;
    const uint32_t n = make_window(pl, lo, hi).n;
    // flatten; count followers per root
    for (uint32_t i = threadIdx.x; i < n; i += CL_BLOCK) { count[i] = 0; fill[i] = 0; }
    __syncthreads();
    uint32_t nf = 0;
    for (uint32_t i = threadIdx.x; i < n; i += CL_BLOCK) {
        uint32_t r = i;
        for (;;) { uint32_t pr = parent[r]; if (pr == r) break; r = pr; }
        if (r != i) { atomicAdd(&count[r], 1u); nf++; }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += CL_BLOCK) {   // second sweep writes the flat roots (after all reads of the tree)
        uint32_t r = i;
        for (;;) { uint32_t pr = parent[r]; if (pr == r) break; r = pr; }
        members[n + i] = r;   // scratch: flat root of i, stored behind the member lists
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += CL_BLOCK) parent[i] = members[n + i];
    __syncthreads();
    // exclusive scan of count -> offset
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    uint32_t mx = 0;
    for (uint32_t base = 0; base < n; base += CL_BLOCK) {
        uint32_t i = base + threadIdx.x;
        uint32_t c = i < n ? count[i] : 0;
        mx = c > mx ? c : mx;
        sh[threadIdx.x] = c;
        __syncthreads();
        for (int o = 1; o < CL_BLOCK; o <<= 1) {
            uint32_t t = threadIdx.x >= (unsigned)o ? sh[threadIdx.x - o] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < n) offset[i] = carry + sh[threadIdx.x] - c;
        __syncthreads();
        if (threadIdx.x == CL_BLOCK - 1) carry += sh[threadIdx.x];
        __syncthreads();
    }
    // scatter followers (unordered inside a cluster; the leader sorts its own short list)
    for (uint32_t i = threadIdx.x; i < n; i += CL_BLOCK) {
        uint32_t r = parent[i];
        if (r != i) members[offset[r] + atomicAdd(&fill[r], 1u)] = i;
    }
    if (nf) atomicAdd(&cnt->followers, (unsigned long long)nf);
    if (mx) atomicMax(&cnt->max_cluster, (unsigned long long)(mx + 1));
}

// ---- D: the walk ---------------------------------------------------------------------------------------
struct WalkCtx {
    Planes pl;
    FdParams fp;
    JTable jt;
    DevCounters* cnt;
    // per-thread accumulators
    unsigned long long nb_processed, nb_skipped, nb_jcheck, nb_no_juncs, n_created;
    // oriented keys this thread's cluster has created in the current window: the snapshot planes of phase A cannot
    // know them, every later in-map test of the cluster has to (tandem repeats inside a piece; later pieces of the cluster)
    static constexpr int NC = 16;
    uint64_t ckey[NC];
    int nc;
    bool c_overflow;    // more than NC creations: fall back to live table lookups
};

// number of set bits of `plane` at positions [a, b)
__device__ __forceinline__ uint32_t range_popc(const uint64_t* plane, uint64_t a, uint64_t b) {
    uint32_t s = 0;
    while (a < b) {
        uint64_t v = fd_bits_at(plane, a);
        uint64_t n = b - a;
        if (n < 64) v &= (1ULL << n) - 1;
        s += (uint32_t)__popcll(v);
        a += 64;
    }
    return s;
}

// NbJCheckKmer increments of the half-steps t in [t0, t1) of the piece whose first window is p0
__device__ __forceinline__ uint32_t jcheck_sum(const Planes& pl, uint64_t p0, int t0, int t1) {
    if (t1 <= t0) return 0;
    // backward-facing half-steps 2q, forward-facing 2q+1
    uint64_t bq0 = (uint64_t)((t0 + 1) >> 1), bq1 = (uint64_t)((t1 + 1) >> 1);
    uint64_t fq0 = (uint64_t)(t0 >> 1), fq1 = (uint64_t)(t1 >> 1);
    return range_popc(pl.cb0, p0 + bq0, p0 + bq1) + 2 * range_popc(pl.cb1, p0 + bq0, p0 + bq1) +
           range_popc(pl.cf0, p0 + fq0, p0 + fq1) + 2 * range_popc(pl.cf1, p0 + fq0, p0 + fq1);
}

// 64 in-map bits for windows [q, q+64) of the piece: the snapshot planes of phase A, plus whatever this thread's
// cluster has created since (compared k-mer by k-mer; no memory traffic), or live lookups once that list overflowed
__device__ void inmap_chunk(const WalkCtx& wc, uint64_t p0, uint32_t q, uint32_t nwin, uint64_t& mF, uint64_t& mB) {
    if (!wc.c_overflow) {
        mF = fd_bits_at(wc.pl.inF, p0 + q);
        mB = fd_bits_at(wc.pl.inB, p0 + q);
        if (wc.nc == 0) return;
        for (uint32_t i = 0; i < 64 && q + i < nwin; i++) {
            uint64_t km = fd_kmer_at(wc.pl.codes, p0 + q + i, wc.fp.k);
            uint64_t rc = fd_revcomp(km, wc.fp.k);
            for (int c = 0; c < wc.nc; c++) {
                if (wc.ckey[c] == km) mF |= 1ULL << i;   // forward-facing key = the k-mer itself
                if (wc.ckey[c] == rc) mB |= 1ULL << i;   // backward-facing key = its reverse complement
            }
        }
        return;
    }
    mF = mB = 0;
    for (uint32_t i = 0; i < 64 && q + i < nwin; i++) {
        uint64_t km = fd_kmer_at(wc.pl.codes, p0 + q + i, wc.fp.k);
        uint64_t rc = fd_revcomp(km, wc.fp.k);
        uint64_t canon = km < rc ? km : rc;
        uint64_t slot;
        uint32_t present;
        if (jt_find_live(wc.jt, canon, slot, present)) {
            if ((present >> (km == canon ? 0 : 1)) & 1u) mF |= 1ULL << i;
            if ((present >> (rc == canon ? 0 : 1)) & 1u) mB |= 1ULL << i;
        }
    }
}

struct JRef {   // a junction record in the table
    uint8_t* rec;
};

__device__ __forceinline__ void rec_update(uint8_t* rec, int idx, int length) {   // Junction::update, Junction.cpp:69-71
    uint8_t l = (uint8_t)length;
    if (rec[idx] < l) rec[idx] = l;
}
__device__ __forceinline__ void rec_add_cov(uint8_t* rec, int nuc) {               // Junction::addCoverage, Junction.cpp:59-67
    uint8_t c = (uint8_t)(rec[5 + nuc] + 1);
    rec[5 + nuc] = c == 0 ? 255 : c;
}

// find or create the junction keyed by the oriented k-mer `key`
__device__ uint8_t* junction_get(WalkCtx& wc, uint64_t key, uint64_t stamp) {
    uint64_t rc = fd_revcomp(key, wc.fp.k);
    uint64_t canon = key < rc ? key : rc;
    int orient = key == canon ? 0 : 1;
    uint64_t slot;
    uint32_t present;
    if (!jt_find_or_claim(wc.jt, canon, slot, present, wc.cnt)) {
        atomicOr(&wc.cnt->error_flags, 1ULL);
        return nullptr;
    }
    uint8_t* rec = wc.jt.recs + (slot * 2 + orient) * 16;
    if (!((present >> orient) & 1u)) {   // JunctionMap::createJunction, JunctionMap.cpp:567-570
        if (wc.nc < WalkCtx::NC) wc.ckey[wc.nc++] = key;
        else wc.c_overflow = true;
        uint64_t* r64 = (uint64_t*)rec;
        r64[0] = 0;
        r64[1] = 0;
        wc.jt.stamps[slot * 2 + orient] = stamp;
        atomicOr((unsigned long long*)&wc.jt.keys[slot], 1ULL << (62 + orient));
        wc.n_created++;
    }
    return rec;
}

// scan_forward (ReadScanner.cpp:112-206) for the piece {p0, nwin}
__device__ void walk_piece(WalkCtx& wc, uint64_t p0, uint32_t nwin, uint64_t piece_seq) {
    const int k = wc.fp.k, j = wc.fp.j;
    const int tmax = 2 * (int)nwin - 2 - 2 * j;     // last half-step with distToEnd > 2j
    const int spacer = 2 * wc.fp.max_spacer - 1;
    int t = 2 * j + 1;
    int last_pos = 0;                               // lastJuncPos
    bool have_last = false;
    uint8_t* last_rec = nullptr;
    int last_t = 0, last_ext_fwd = 0;

    while (t <= tmax) {
        // ---- find_next_junction (ReadScanner.cpp:61-86): first t' >= t that is in the map, hits the spacer rule, or is flagged
        int t_sp = last_pos + spacer;
        if (t_sp < t) t_sp = t;
        int t_ev = 0x7fffffff;
        {
            uint32_t q0 = (uint32_t)(t >> 1);
            for (uint32_t qc = q0; qc < nwin && 2 * (int)qc <= tmax && 2 * (int)qc <= t_sp; qc += 64) {
                uint64_t mF, mB;
                inmap_chunk(wc, p0, qc, nwin, mF, mB);
                uint64_t eF = mF | fd_bits_at(wc.pl.ff, p0 + qc);
                uint64_t eB = mB | fd_bits_at(wc.pl.fb, p0 + qc);
                if (qc == q0 && (t & 1)) eB &= ~1ULL;   // the backward-facing half-step of q0 is already behind us
                uint32_t rem = nwin - qc;
                if (rem < 64) { uint64_t m = (1ULL << rem) - 1; eF &= m; eB &= m; }
                int tb = eB ? 2 * (int)(qc + __builtin_ctzll(eB)) : 0x7fffffff;
                int tf = eF ? 2 * (int)(qc + __builtin_ctzll(eF)) + 1 : 0x7fffffff;
                int te = tb < tf ? tb : tf;
                if (te != 0x7fffffff) { t_ev = te; break; }
            }
        }
        int tn = t_ev < t_sp ? t_ev : t_sp;
        if (tn > tmax) {   // ran off the end of the piece
            wc.nb_processed += (unsigned long long)(tmax - t + 1);
            wc.nb_jcheck += jcheck_sum(wc.pl, p0, t, tmax + 1);
            break;
        }
        const uint32_t q = (uint32_t)(tn >> 1);
        const bool fwd = tn & 1;
        // why did we stop here?  (order of the tests in find_next_junction)
        bool in_map;
        {
            uint64_t mF, mB;
            inmap_chunk(wc, p0, q, nwin, mF, mB);
            in_map = (fwd ? mF : mB) & 1ULL;
        }
        const bool by_spacer = !in_map && (tn - last_pos >= spacer);
        wc.nb_processed += (unsigned long long)(tn - t);
        wc.nb_jcheck += jcheck_sum(wc.pl, p0, t, (in_map || by_spacer) ? tn : tn + 1);

        // ---- junction at (q, fwd)  (ReadScanner.cpp:133-192)
        uint64_t km = fd_kmer_at(wc.pl.codes, p0 + q, k);
        uint64_t key = fwd ? km : fd_revcomp(km, k);
        int real = fwd ? fd_base_at(wc.pl.codes, p0 + q + k) : (fd_base_at(wc.pl.codes, p0 + q - 1) ^ 2);
        uint8_t* rec = junction_get(wc, key, (piece_seq << 16) | (uint64_t)tn);
        if (!rec) return;
        last_pos = tn;
        rec_add_cov(rec, real);
        const int ext_fwd = fwd ? real : 4;          // getExtensionIndex(FORWARD)
        const int ext_bwd = fwd ? 4 : real;          // getExtensionIndex(BACKWARD)
        if (have_last) {                             // directLinkJunctions, JunctionMap.cpp:551-561
            int d = tn - last_t;
            rec_update(last_rec, last_ext_fwd, d);
            rec_update(rec, ext_bwd, d);
            last_rec[9] |= (uint8_t)(1u << last_ext_fwd);
            rec[9] |= (uint8_t)(1u << ext_bwd);
        } else {
            have_last = true;
            rec_update(rec, ext_bwd, tn - 2 * j);
        }
        last_rec = rec;
        last_t = tn;
        last_ext_fwd = ext_fwd;
        int d = rec[ext_fwd];
        if (d < 1) d = 1;
        t = tn + d;
        wc.nb_processed += 1;
        wc.nb_skipped += (unsigned long long)(d - 1);
    }

    if (!have_last) {   // add_fake_junction (ReadScanner.cpp:92-104)
        wc.nb_no_juncs++;
        const int len = (int)nwin + k - 1;
        const int m = len / 2 - k / 2;
        uint64_t key = fd_kmer_at(wc.pl.codes, p0 + m, k);
        int real = fd_base_at(wc.pl.codes, p0 + m + k);
        uint8_t* rec = junction_get(wc, key, (piece_seq << 16) | 0xFFFFULL);
        if (!rec) return;
        rec_add_cov(rec, real);
        const int tm = 2 * m + 1;
        rec_update(rec, 4, tm - 2 * j);
        rec_update(rec, real, (2 * (int)nwin - 1 - tm) - 2 * j);
    } else {            // ReadScanner.cpp:202-206
        rec_update(last_rec, last_ext_fwd, (2 * (int)nwin - 1 - last_t) - 2 * j);
    }
}

__global__ void __launch_bounds__(64) k_walk(Planes pl, FdParams fp, JTable jt, const uint32_t* __restrict__ root,
                                             const uint32_t* __restrict__ count, const uint32_t* __restrict__ offset, uint32_t* members,
                                             uint64_t lo, uint64_t hi, uint64_t piece_seq_base, DevCounters* cnt) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    WalkCtx wc;
    wc.pl = pl; wc.fp = fp; wc.jt = jt; wc.cnt = cnt;
    wc.nb_processed = wc.nb_skipped = wc.nb_jcheck = wc.nb_no_juncs = wc.n_created = 0;
    wc.nc = 0; wc.c_overflow = false;
    const WinDesc wd = make_window(pl, lo, hi);
    const uint32_t n = wd.n, first_piece = wd.first_piece;
    if (i < n && root[i] == i) {
        uint2 pc = pl.pieces[first_piece + i];
        walk_piece(wc, pc.x, pc.y, piece_seq_base + first_piece + i);
        uint32_t nm = count[i];
        if (nm) {
            uint32_t* mem = members + offset[i];
            for (uint32_t a = 1; a < nm; a++) {   // insertion sort: ascending piece order
                uint32_t v = mem[a];
                uint32_t b = a;
                while (b > 0 && mem[b - 1] > v) { mem[b] = mem[b - 1]; b--; }
                mem[b] = v;
            }
            for (uint32_t a = 0; a < nm; a++) {
                uint32_t m = mem[a];
                uint2 pm = pl.pieces[first_piece + m];
                walk_piece(wc, pm.x, pm.y, piece_seq_base + first_piece + m);
            }
        }
    }
    // wave-level reduction of the counters
    unsigned long long v[5] = {wc.nb_processed, wc.nb_skipped, wc.nb_jcheck, wc.nb_no_juncs, wc.n_created};
    for (int c = 0; c < 5; c++)
        for (int o = 32; o > 0; o >>= 1) v[c] += __shfl_down(v[c], o, 64);
    if (fd_lane() == 0) {
        if (v[0]) atomicAdd(&cnt->nb_processed, v[0]);
        if (v[1]) atomicAdd(&cnt->nb_skipped, v[1]);
        if (v[2]) atomicAdd(&cnt->nb_jcheck, v[2]);
        if (v[3]) atomicAdd(&cnt->nb_no_juncs, v[3]);
        if (v[4]) atomicAdd(&cnt->n_junctions, v[4]);
    }
}

// ---- E: sparse reset of the window table ----------------------------------------------------------------
// The two "slots used" counters alternate between consecutive windows, so this kernel can also zero the one the
// next window will count into (it was consumed by the previous window's clean-up).
__global__ void __launch_bounds__(256) k_walk_clean(WTable wt, DevCounters* cnt, uint32_t* parent, Planes pl, uint64_t lo, uint64_t hi,
                                                    int parity) {
    uint64_t used = parity ? cnt->wt_used_b : cnt->wt_used;
    const uint32_t n = make_window(pl, lo, hi).n;
    if (blockIdx.x == 0 && threadIdx.x == 0) { if (parity) cnt->wt_used = 0; else cnt->wt_used_b = 0; }
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t a = i; a < used; a += stride) {
        uint32_t s = wt.slots[a];
        uint64_t key = wt.keys[s];
        uint32_t b = (uint32_t)(fd_mix(key) >> 40) & ((1u << WBITS_LOG2) - 1);
        wt.bits[b >> 5] = 0;   // whole word: every bit of it belongs to a key that is being removed as well
        wt.keys[s] = J_EMPTY;
        wt.owner[s] = U_INF;
    }
    for (uint64_t a = i; a < n; a += stride) parent[a] = (uint32_t)a;
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

// ---- export: compact the present records ------------------------------------------------------------------
struct ExportEntry {   // FGPU_TABLE_ENTRY_BYTES = 32
    uint64_t key;      // oriented k-mer
    uint64_t stamp;
    uint8_t rec[16];
};

__global__ void __launch_bounds__(256) k_export(JTable jt, FdParams fp, ExportEntry* out, uint64_t* stamps_out, unsigned long long* n_out) {
    uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s > jt.mask) return;
    uint64_t w = jt.keys[s];
    if (w == J_EMPTY) return;
    uint64_t canon = w & J_KEYMASK;
    for (int o = 0; o < 2; o++) {
        if (!((w >> (62 + o)) & 1ULL)) continue;
        unsigned long long idx = atomicAdd(n_out, 1ULL);
        ExportEntry e;
        e.key = o == 0 ? canon : fd_revcomp(canon, fp.k);
        e.stamp = jt.stamps[s * 2 + o];
        const uint64_t* r = (const uint64_t*)(jt.recs + (s * 2 + o) * 16);
        ((uint64_t*)e.rec)[0] = r[0];
        ((uint64_t*)e.rec)[1] = r[1];
        out[idx] = e;
        stamps_out[idx] = e.stamp;
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

__global__ void __launch_bounds__(256) k_import(JTable jt, FdParams fp, const ExportEntry* in, uint64_t n, DevCounters* cnt) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ExportEntry e = in[i];
    uint64_t rc = fd_revcomp(e.key, fp.k);
    uint64_t canon = e.key < rc ? e.key : rc;
    int orient = e.key == canon ? 0 : 1;
    uint64_t slot;
    uint32_t present;
    if (!jt_find_or_claim(jt, canon, slot, present, cnt)) { atomicOr(&cnt->error_flags, 1ULL); return; }
    uint64_t* r = (uint64_t*)(jt.recs + (slot * 2 + orient) * 16);
    r[0] = ((const uint64_t*)e.rec)[0];
    r[1] = ((const uint64_t*)e.rec)[1];
    jt.stamps[slot * 2 + orient] = e.stamp;
    atomicOr((unsigned long long*)&jt.keys[slot], 1ULL << (62 + orient));
}

JTable make_jt(fgpu_ctx* ctx) { return JTable{ctx->jkeys, ctx->jrecs, ctx->jstamps, ctx->jcap - 1}; }
WTable make_wt(fgpu_ctx* ctx) { return WTable{ctx->wkeys, ctx->wowner, ctx->wslots, ctx->wbits, ctx->wcap - 1}; }

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
int fgpu_scan_alloc(fgpu_ctx* ctx) {
    if (ctx->jkeys) return FGPU_OK;
    ctx->jcap = ctx->prm.junction_capacity;
    FGPU_HIP(hipMalloc(&ctx->jkeys, ctx->jcap * 8));
    FGPU_HIP(hipMalloc(&ctx->jrecs, ctx->jcap * 32));
    FGPU_HIP(hipMalloc(&ctx->jstamps, ctx->jcap * 16));
    // Scheduling windows span at most FGPU_MAX_SPAN positions (+ one piece length).  Worst case every position is a
    // candidate with a distinct k-mer, so the window table holds 2x that; piece starts are >= k+1 apart.
    ctx->wcap = 4 * FGPU_MAX_SPAN;
    ctx->wmax = (uint32_t)(FGPU_MAX_SPAN / (uint64_t)(ctx->fd.k + 1) + 2);
    FGPU_HIP(hipMalloc(&ctx->wdesc, 64));
    FGPU_HIP(hipMalloc(&ctx->wkeys, ctx->wcap * 8));
    FGPU_HIP(hipMalloc(&ctx->wowner, ctx->wcap * 4));
    FGPU_HIP(hipMalloc(&ctx->wslots, ctx->wcap * 4));
    FGPU_HIP(hipMalloc(&ctx->wbits, (1ULL << WBITS_LOG2) / 8));
    FGPU_HIP(hipMalloc(&ctx->uf_parent, ctx->wmax * 4));
    FGPU_HIP(hipMalloc(&ctx->cl_count, ctx->wmax * 4));
    FGPU_HIP(hipMalloc(&ctx->cl_offset, ctx->wmax * 4));
    FGPU_HIP(hipMalloc(&ctx->cl_fill, ctx->wmax * 4));
    FGPU_HIP(hipMalloc(&ctx->cl_members, ctx->wmax * 4 * 2));
    return FGPU_OK;
}

int fgpu_scan_reset(fgpu_ctx* ctx) {
    FGPU_HIP(hipMemsetAsync(ctx->jkeys, 0xFF, ctx->jcap * 8, ctx->stream));
    FGPU_HIP(hipMemsetAsync(ctx->wkeys, 0xFF, ctx->wcap * 8, ctx->stream));
    FGPU_HIP(hipMemsetAsync(ctx->wowner, 0xFF, ctx->wcap * 4, ctx->stream));
    FGPU_HIP(hipMemsetAsync(ctx->wbits, 0, (1ULL << WBITS_LOG2) / 8, ctx->stream));
    FGPU_LAUNCH("iota", k_iota_u32, 64, 256, ctx->uf_parent, (uint64_t)ctx->wmax);
    return FGPU_OK;
}

// Walk the pieces of the current batch, scheduling window after scheduling window (position ranges of
// ctx->window_span stream positions).  No host round trip: window extents are derived on the device.
int fgpu_stage_scan_walk(fgpu_ctx* ctx, uint64_t n_pieces) {
    if (!n_pieces) return FGPU_OK;
    BatchBufs& bb = *ctx->cur;
    Planes pl{(const uint64_t*)bb.codes.p, (const uint64_t*)bb.pm.p, (const uint64_t*)bb.ps.p, (const uint32_t*)bb.ps_prefix.p,
              (const uint64_t*)bb.ff.p, (const uint64_t*)bb.fb.p, (const uint64_t*)bb.cf0.p, (const uint64_t*)bb.cf1.p,
              (const uint64_t*)bb.cb0.p, (const uint64_t*)bb.cb1.p, (uint64_t*)bb.inF.p, (uint64_t*)bb.inB.p, (const uint2*)bb.pieces.p};
    JTable jt = make_jt(ctx);
    WTable wt = make_wt(ctx);
    const uint64_t span = ctx->window_span;
    const uint64_t ext = bb.max_piece_span;          // a piece that starts inside the window may reach this far beyond it
    const uint64_t T = bb.T;
    const uint64_t seq_base = ctx->scan_piece_base;
    const unsigned walk_grid = fgpu_blocks(ctx->wmax, 64);
    // thousands of tiny launches: by default one event pair around the whole stage
    const int stage_tok = fgpu_prof_begin(ctx, "walk_stage");
    ctx->prof_suppress = !ctx->prof_walk_detail;
    for (uint64_t lo = 0; lo < T; lo += span) {
        const uint64_t hi = std::min<uint64_t>(T, lo + span);
        const uint64_t pos_end = std::min<uint64_t>(T, hi + ext);
        const unsigned grid = fgpu_blocks((pos_end - (lo & ~63ULL) + 63) & ~63ULL, 256);
        const int parity = (int)(ctx->scan_windows & 1);
        FGPU_LAUNCH("walk_lookup", k_walk_lookup, grid, 256, pl, ctx->fd, jt, wt, ctx->uf_parent, lo, hi, pos_end, ctx->counters, parity);
        FGPU_LAUNCH("walk_link", k_walk_link, grid, 256, pl, ctx->fd, wt, ctx->uf_parent, lo, hi, pos_end);
        FGPU_LAUNCH("walk_cluster", k_walk_cluster, 1, CL_BLOCK, ctx->uf_parent, ctx->cl_count, ctx->cl_offset, ctx->cl_fill,
                    ctx->cl_members, pl, lo, hi, ctx->counters);
        FGPU_LAUNCH("walk", k_walk, walk_grid, 64, pl, ctx->fd, jt, (const uint32_t*)ctx->uf_parent, (const uint32_t*)ctx->cl_count,
                    (const uint32_t*)ctx->cl_offset, ctx->cl_members, lo, hi, seq_base, ctx->counters);
        FGPU_LAUNCH("walk_clean", k_walk_clean, 64, 256, wt, ctx->counters, ctx->uf_parent, pl, lo, hi, parity);
        ctx->scan_windows++;
    }
    ctx->prof_suppress = false;
    fgpu_prof_end(ctx, stage_tok);
    ctx->scan_piece_base += n_pieces;
    return FGPU_OK;
}

// ---- junction download / export / import -------------------------------------------------------------------------
int fgpu_scan_export_impl(fgpu_ctx* ctx, void* dev_entries, uint64_t cap_entries, uint64_t* d_stamps, uint64_t* n_entries) {
    unsigned long long* d_n = &ctx->counters->pad;
    FGPU_HIP(hipMemsetAsync(d_n, 0, 8, ctx->stream));
    (void)cap_entries;
    FGPU_LAUNCH("export", k_export, fgpu_blocks(ctx->jcap, 256), 256, make_jt(ctx), ctx->fd, (ExportEntry*)dev_entries, d_stamps, d_n);
    FGPU_HIP(hipMemcpyAsync(&ctx->counters_host->pad, d_n, 8, hipMemcpyDeviceToHost, ctx->stream));
    FGPU_HIP(hipStreamSynchronize(ctx->stream));
    *n_entries = ctx->counters_host->pad;
    return FGPU_OK;
}

int fgpu_scan_import_impl(fgpu_ctx* ctx, const void* dev_entries, uint64_t n) {
    if (!n) return FGPU_OK;
    FGPU_LAUNCH("import", k_import, fgpu_blocks(n, 256), 256, make_jt(ctx), ctx->fd, (const ExportEntry*)dev_entries, n, ctx->counters);
    return FGPU_OK;
}

// creation-ordered download: export, radix-sort the stamps (rocPRIM; not a hot step), gather
int fgpu_scan_download_impl(fgpu_ctx* ctx, uint64_t* keys_host, fgpu_junction* recs_host, uint64_t cap, uint64_t* n_out) {
    uint64_t n_max = ctx->scan_stats.n_junctions;
    FGPU_HIP(hipMemcpyAsync(&ctx->counters_host->n_junctions, &ctx->counters->n_junctions, 8, hipMemcpyDeviceToHost, ctx->stream));
    FGPU_HIP(hipStreamSynchronize(ctx->stream));
    n_max = ctx->counters_host->n_junctions + ctx->scan_imported;
    *n_out = n_max;
    if (!n_max || !keys_host || !recs_host) return FGPU_OK;
    if (cap < n_max) { ctx->err = "junction buffer too small"; return FGPU_ERR_CAPACITY; }
    ExportEntry* d_entries = nullptr;
    uint64_t *d_stamps = nullptr, *d_stamps_sorted = nullptr, *d_keys = nullptr;
    uint32_t *d_idx = nullptr, *d_idx_sorted = nullptr;
    fgpu_junction* d_recs = nullptr;
    void* d_tmp = nullptr;
    int rc = FGPU_OK;
    hipError_t e;
#define DL_HIP(call) do { e = (call); if (e != hipSuccess) { ctx->err = std::string(#call) + ": " + hipGetErrorString(e); rc = FGPU_ERR_HIP; goto done; } } while (0)
    DL_HIP(hipMalloc(&d_entries, n_max * sizeof(ExportEntry)));
    DL_HIP(hipMalloc(&d_stamps, n_max * 8));
    DL_HIP(hipMalloc(&d_stamps_sorted, n_max * 8));
    DL_HIP(hipMalloc(&d_idx, n_max * 4));
    DL_HIP(hipMalloc(&d_idx_sorted, n_max * 4));
    DL_HIP(hipMalloc(&d_keys, n_max * 8));
    DL_HIP(hipMalloc(&d_recs, n_max * sizeof(fgpu_junction)));
    {
        uint64_t n = 0;
        rc = fgpu_scan_export_impl(ctx, d_entries, n_max, d_stamps, &n);
        if (rc) goto done;
        if (n != n_max) { ctx->err = "junction count mismatch between counters and table"; rc = FGPU_ERR_STATE; goto done; }
        hipLaunchKernelGGL(k_iota_u32, dim3(256), dim3(256), 0, ctx->stream, d_idx, n);
        size_t tmp_bytes = 0;
        DL_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_stamps, d_stamps_sorted, d_idx, d_idx_sorted, n, 0, 64, ctx->stream));
        DL_HIP(hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 16));
        DL_HIP(rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_stamps, d_stamps_sorted, d_idx, d_idx_sorted, n, 0, 64, ctx->stream));
        hipLaunchKernelGGL(k_gather_sorted, dim3(fgpu_blocks(n, 256)), dim3(256), 0, ctx->stream, (const ExportEntry*)d_entries,
                           (const uint32_t*)d_idx_sorted, n, d_keys, d_recs);
        DL_HIP(hipMemcpyAsync(keys_host, d_keys, n * 8, hipMemcpyDeviceToHost, ctx->stream));
        DL_HIP(hipMemcpyAsync(recs_host, d_recs, n * sizeof(fgpu_junction), hipMemcpyDeviceToHost, ctx->stream));
        DL_HIP(hipStreamSynchronize(ctx->stream));
    }
done:
#undef DL_HIP
    hipFree(d_entries); hipFree(d_stamps); hipFree(d_stamps_sorted); hipFree(d_idx); hipFree(d_idx_sorted);
    hipFree(d_keys); hipFree(d_recs); hipFree(d_tmp);
    return rc;
}
