// fgpu_device.h — device-side k-mer codec, oldHash and Bloom bit access for gfx950.
//
// Semantics restated from the reference (behaviour, not text):
//   2-bit code A0 C1 T2 G3 = (ascii >> 1) & 3                      utils/Kmer.cpp:82-88
//   k-mer = 2k low bits, first base most significant                 utils/Kmer.cpp:410-433
//   reverse complement, canonical = min(x, rc(x))                    utils/Kmer.cpp:238-252,531-533
//   oldHash with seed_tab[0], seed_tab[1] (user_seed = 0)            utils/Bloom.h:134-145, Bloom.cpp:500-511
//   bit i of an element = (hA + i*hB) mod tai, byte p>>3, mask 1<<(p&7)   utils/Bloom.h:217-226,242-258
//
// Internal stream layout (built by pack.hip, see DESIGN.md "data layout in HBM"):
//   codes[w]  : 32 bases per 64-bit word, first base in the two MOST significant bits
//   bad[w]    : 1 bit per stream position (LSB = lowest position); 1 = not ACGT / read separator / past the end
// so a k-mer window is one funnel shift of two code words and its validity one funnel shift of two
// mask words.  All other per-position planes use the same LSB-first bit order as `bad`.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FD_SEED0 0xffaa54ffe6e6e6e7ULL   // seed_tab[0]
#define FD_SEED1 0x1140aada557088a4ULL   // seed_tab[1]

struct FdParams {
    int      k;
    int      j;
    int      n_hash;
    int      max_spacer;
    uint64_t kmask;      // (1 << 2k) - 1
    uint64_t tai_mask;   // tai - 1
};

__device__ __forceinline__ uint64_t fd_revcomp(uint64_t x, int k) {
    // complement each 2-bit group (x ^ 2), reverse the groups, right-align
    uint64_t y = __builtin_bitreverse64(x ^ 0xAAAAAAAAAAAAAAAAULL);
    y = ((y >> 1) & 0x5555555555555555ULL) | ((y & 0x5555555555555555ULL) << 1);
    return y >> (64 - 2 * k);
}

__device__ __forceinline__ uint64_t fd_canon(uint64_t x, int k) {
    uint64_t r = fd_revcomp(x, k);
    return x < r ? x : r;
}

__device__ __forceinline__ uint64_t fd_old_hash(uint64_t key, uint64_t seed) {
    uint64_t h = seed;
    h ^= (h << 7) ^ (key * (h >> 3)) ^ (~((h << 11) + (key ^ (h >> 5))));
    h = (~h) + (h << 21);
    h ^= h >> 24;
    h = (h + (h << 3)) + (h << 8);
    h ^= h >> 14;
    h = (h + (h << 2)) + (h << 4);
    h ^= h >> 28;
    h += h << 31;
    return h;
}

// The seed-dependent constants of the first mixing line fold at compile time:
//   h ^= (h<<7) ^ key*(h>>3) ^ ~((h<<11) + (key ^ (h>>5)))
__device__ __forceinline__ void fd_hash_pair(uint64_t canon, uint64_t tai_mask, uint64_t& hA, uint64_t& hB) {
    hA = fd_old_hash(canon, FD_SEED0) & tai_mask;
    hB = fd_old_hash(canon, FD_SEED1) & tai_mask;
}

// k-mer whose first base is stream position p
__device__ __forceinline__ uint64_t fd_kmer_at(const uint64_t* __restrict__ codes, uint64_t p, int k) {
    uint64_t w = p >> 5;
    int o = (int)(p & 31) * 2;
    uint64_t hi = codes[w], lo = codes[w + 1];
    uint64_t v = (hi << o) | ((lo >> 1) >> (63 - o));   // branch-free funnel (o may be 0): both loads issue together
    return v >> (64 - 2 * k);
}

__device__ __forceinline__ int fd_base_at(const uint64_t* __restrict__ codes, uint64_t p) {
    return (int)((codes[p >> 5] >> (62 - 2 * (int)(p & 31))) & 3);
}

// 64 plane bits starting at position p (bit 0 = position p)
__device__ __forceinline__ uint64_t fd_bits_at(const uint64_t* __restrict__ plane, uint64_t p) {
    uint64_t w = p >> 6;
    int o = (int)(p & 63);
    uint64_t lo = plane[w], hi = plane[w + 1];
    return (lo >> o) | ((hi << 1) << (63 - o));          // branch-free funnel (o may be 0): both loads issue together
}

// window [p, p+k) free of bad positions?
__device__ __forceinline__ bool fd_window_ok(const uint64_t* __restrict__ bad, uint64_t p, int k) {
    uint64_t v = fd_bits_at(bad, p);
    return (v & ((1ULL << k) - 1)) == 0;
}

// Bloom::contains with the reference's early exit (utils/Bloom.h:242-258); filter viewed as 32-bit
// little-endian words, so bit p of the byte array is bit (p & 31) of word p >> 5.
__device__ __forceinline__ bool fd_bloom_contains(const uint32_t* __restrict__ bloom, uint64_t hA, uint64_t hB,
                                                  uint64_t tai_mask, int n_hash) {
    uint64_t h = hA;
    for (int i = 0; i < n_hash; i++) {
        if (!((bloom[h >> 5] >> (h & 31)) & 1u)) return false;
        h = (h + hB) & tai_mask;
    }
    return true;
}

__device__ __forceinline__ bool fd_bloom_contains_canon(const uint32_t* __restrict__ bloom, uint64_t canon,
                                                        uint64_t tai_mask, int n_hash) {
    uint64_t hA, hB;
    fd_hash_pair(canon, tai_mask, hA, hB);
    return fd_bloom_contains(bloom, hA, hB, tai_mask, n_hash);
}

// The same answer for k-mers that are mostly ABSENT (alternate extensions): the second hash is only computed when the
// first bit is set, which is the exception there; oldHash costs more instructions than the probe it feeds.
__device__ __forceinline__ bool fd_bloom_contains_canon_lazy(const uint32_t* __restrict__ bloom, uint64_t canon,
                                                             uint64_t tai_mask, int n_hash) {
    const uint64_t hA = fd_old_hash(canon, FD_SEED0) & tai_mask;
    if (!((bloom[hA >> 5] >> (hA & 31)) & 1u)) return false;
    if (n_hash == 1) return true;
    const uint64_t hB = fd_old_hash(canon, FD_SEED1) & tai_mask;
    uint64_t h = (hA + hB) & tai_mask;
    for (int i = 1; i < n_hash; i++) {
        if (!((bloom[h >> 5] >> (h & 31)) & 1u)) return false;
        h = (h + hB) & tai_mask;
    }
    return true;
}

// Bloom::add (utils/Bloom.h:217-226) on a monotone bitmap shared by the whole grid: test first
// (a stale 0 only costs a redundant atomic; a 1 is never stale because bits are never cleared).
__device__ __forceinline__ void fd_bloom_set(uint32_t* bloom, uint64_t hA, uint64_t hB, uint64_t tai_mask, int n_hash) {
    uint64_t h = hA;
    for (int i = 0; i < n_hash; i++) {
        uint32_t bit = 1u << (h & 31);
        if (!(bloom[h >> 5] & bit)) atomicOr(&bloom[h >> 5], bit);
        h = (h + hB) & tai_mask;
    }
}

// 64-bit mix for the junction / window hash tables (not part of any on-disk format)
__device__ __forceinline__ uint64_t fd_mix(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}

__device__ __forceinline__ int fd_lane() { return (int)(threadIdx.x & 63); }
