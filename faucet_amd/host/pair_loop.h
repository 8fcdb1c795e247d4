// pair_loop.h — scanReads' paired-end loop on the HOST, over the lists the device hands out: the hosts' way on when the device cannot hold the
// long pair filter's working state.
//
// fgpu_scan_long_pairs keeps 4 bytes of HBM per filter bit (first-set times of its fixed point, csrc/pairs.hip); `--high_cov` sizes that filter
// at estimated_kmers / 2 x 9 bits (src/Faucet.cpp:279-280), so a large enough run gets FGPU_ERR_NOMEM there -- where the reference has no
// limit but host memory.  The hosts (faucet_main.cpp, integration/faucet_binding.cpp) then run the reference's own loop
// (src/ReadScanner.cpp:304-351) over scanInputRead's lists as fgpu_scan_take_stops returns them, batch by batch, in file order:
//     first end:  back1 = its list          second end:  both lists non-empty -> not_empty_count++, and with cleaning on, for every element of
//     back1: look for a partner among back2's with containsPair; if none, addPair(element, back2.front())   else empty_count++
// into a filter in host memory with the format's bit positions (Bloom::addPair / containsPair, utils/Bloom.cpp:127-154: the two canonical
// k-mers, the smaller hashed with seed_tab[0] and the larger with seed_tab[1], bit i at (h0 + i * h1) mod tai; byte p >> 3, mask 1 << (p & 7)).
// Host logic over the C ABI's plain structs: no HIP here.  
#pragma once
#include <stdint.h>

#include <vector>

#include "faucet_gpu.h"

namespace faucet_host {

class HostLongPairs {
public:
    // bits: tai / 8 bytes owned by the caller (may be NULL when filter == false: --no_cleaning only counts)
    HostLongPairs(uint8_t* bits, uint64_t tai, int n_hash, int k, bool filter) : bits_(bits), mask_(tai - 1), n_hash_(n_hash), k_(k), filter_(filter && bits) {}

    uint64_t empty_count = 0, not_empty_count = 0;

    // the lists of one scanned batch (fgpu_scan_take_stops), flattened in file order, and the number of records the batch held
    void batch(const fgpu_stop* stops, uint64_t n_stops, uint64_t n_reads) {
        uint64_t i = 0;
        for (uint64_t r = 0; r < n_reads; r++) {
            cur_.clear();
            while (i < n_stops && stops[i].read == r) cur_.push_back(stops[i++].ext);
            record();
        }
    }

private:
    static uint64_t revcomp(uint64_t x, int k) {   // utils/Kmer.cpp:238-252 (A0 C1 T2 G3: the complement of a base is base ^ 2)
        uint64_t r = 0;
        for (int i = 0; i < k; i++) { r = (r << 2) | ((x & 3) ^ 2); x >>= 2; }
        return r;
    }
    static uint64_t old_hash(uint64_t key, uint64_t seed) {   // utils/Bloom.h:134-145
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
    void hashes(uint64_t k1, uint64_t k2, uint64_t* h0, uint64_t* h1) const {
        const uint64_t r1 = revcomp(k1, k_), r2 = revcomp(k2, k_);
        const uint64_t e1 = k1 < r1 ? k1 : r1, e2 = k2 < r2 ? k2 : r2;
        *h0 = old_hash(e1 < e2 ? e1 : e2, 0xffaa54ffe6e6e6e7ULL) & mask_;     // seed_tab[0] (utils/Bloom.cpp:500-511, user_seed 0)
        *h1 = old_hash(e1 < e2 ? e2 : e1, 0x1140aada557088a4ULL) & mask_;     // seed_tab[1]
    }
    bool contains(uint64_t k1, uint64_t k2) const {
        uint64_t h, h1;
        hashes(k1, k2, &h, &h1);
        for (int i = 0; i < n_hash_; i++, h = (h + h1) & mask_)
            if (!(bits_[h >> 3] & (1u << (h & 7)))) return false;
        return true;
    }
    void add(uint64_t k1, uint64_t k2) {
        uint64_t h, h1;
        hashes(k1, k2, &h, &h1);
        for (int i = 0; i < n_hash_; i++, h = (h + h1) & mask_) bits_[h >> 3] |= (uint8_t)(1u << (h & 7));
    }
    void record() {   // the body of scanReads' loop for one record (src/ReadScanner.cpp:304-351)
        if (first_end_) {
            back1_ = cur_;
        } else {
            if (!back1_.empty() && !cur_.empty()) {
                not_empty_count++;
                if (filter_)
                    for (uint64_t pair1 : back1_) {
                        bool paired = false;
                        for (uint64_t pair2 : cur_)
                            if (contains(pair1, pair2)) { paired = true; break; }
                        if (!paired) add(pair1, cur_.front());
                    }
            } else {
                empty_count++;
            }
        }
        first_end_ = !first_end_;
    }

    uint8_t* bits_;
    uint64_t mask_;
    int n_hash_, k_;
    bool filter_;
    bool first_end_ = true;
    std::vector<uint64_t> back1_, cur_;
};

}  // namespace faucet_host
