// The order in which the reference dumps its junctions, without the container.
//
// The reference keeps its junctions in a std::unordered_map<kmer_type, Junction> (utils/JunctionMap.h:61) and JunctionMap::writeToFile
// (utils/JunctionMap.cpp:579-596) walks it from begin() to end(): the `.junctions` file is in the container's iteration order, which is
// a function of the sequence of insertions and nothing else.  The device hands the junctions over in creation order (= the reference's
// insertion order), so filling the same container reproduces the file -- but a million heap nodes, chased one dependent cache miss at a
// time, cost more than both device passes (0.23 s of a 0.9 s run on config 2).  DumpOrder replays what the container does to its node
// list on two flat index arrays (4 bytes per junction: they stay in cache) and yields the same sequence:
//   * libstdc++ keeps ALL nodes on one singly linked list; a bucket points at the node BEFORE its first node;
//   * a node inserted into an empty bucket goes to the front of the whole list, into a non-empty bucket right behind that "before"
//     node (bits/hashtable.h, _M_insert_bucket_begin);
//   * a rehash relinks the nodes in list order by the same two rules (_M_rehash_aux, unique keys);
//   * WHEN it rehashes and to how many buckets is asked of the library's own policy object (std::__detail::_Prime_rehash_policy), not
//     restated here; the hash of a 64-bit integer is the integer.
// Keys must be distinct (the device's are).  `agrees_with_the_container` checks the replay against a real std::unordered_map on a prefix
// of the keys; the CLI runs it before trusting the replay and falls back to the container if it ever fails (another standard library).
//
// Round 6: the replay is one dependent cache miss after the other -- 5.7 s for config 5's 29.5 M junctions, 60 % of the command line's wall time
// there (profiles/r06_cli_large.txt).  But the two rules have a CLOSED FORM.  Between two rehashes the list is: the buckets' groups in the
// order in which the buckets FIRST received a node, latest first; inside a group the nodes latest first.  I.e. insert the sequence S into an
// empty table of B buckets and the list is S sorted by (first time of the node's bucket, descending; own time, descending).  A rehash
// re-inserts the list as it stands into the new buckets by the same rules, so with rehashes at counts c_1 < c_2 < ... (to B_1, B_2, ... buckets)
//     L_j = sorted( L_(j-1) followed by the nodes c_j .. c_(j+1) - 1,  by B_j )        and the dump order is the last L.
// The sizes double, so all the sorts together are about 2 n elements: `schedule` asks the library's policy object WHEN it rehashes and to how
// many buckets (no container, no memory traffic), `of_sorted` does the sorts on the host (the check below and the tests), and the device does
// them for the command line (fgpu_scan_dump_order: radix sorts, milliseconds for 3e7 junctions).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <algorithm>
#include <unordered_map>
#include <vector>

class DumpOrder {
public:
    // order of dump of keys[0..n), as indices into keys
    static std::vector<uint32_t> of(const uint64_t* keys, size_t n) {
        DumpOrder d(keys, n);
        for (size_t i = 0; i < n; i++) d.insert((uint32_t)i);
        std::vector<uint32_t> order;
        order.reserve(n);
        for (uint32_t i = d.head_; i != kNil; i = d.next_[i]) order.push_back(i);
        return order;
    }
    static bool agrees_with_the_container(const uint64_t* keys, size_t n) {
        std::unordered_map<uint64_t, uint32_t> real;
        for (size_t i = 0; i < n; i++) real.insert(std::pair<uint64_t, uint32_t>(keys[i], (uint32_t)i));
        if (real.size() != n) return false;   // repeated keys: not what the replay is for
        const std::vector<uint32_t> order = of(keys, n);
        size_t at = 0;
        for (const auto& kv : real)
            if (at >= order.size() || order[at++] != kv.second) return false;
        return at == order.size() && of_sorted(keys, n) == order;      // ... and the closed form (what the device computes) gives the same
    }

    // WHEN the container rehashes while n keys are inserted one by one, and to how many buckets: {nodes present, buckets from then on}, the
    // first entry being the empty container's {0, 1}.  Asked of the library's own policy object, insertion by insertion as the container asks
    // (bits/hashtable.h, _M_insert_unique_node); a call is a comparison unless it rehashes.
    struct Rehash { uint64_t count, buckets; };
    static std::vector<Rehash> schedule(size_t n) {
        std::__detail::_Prime_rehash_policy policy;
        std::vector<Rehash> out(1, Rehash{0, 1});
        size_t buckets = 1;
        for (size_t count = 0; count < n; count++) {
            const std::pair<bool, size_t> grow = policy._M_need_rehash(buckets, count, 1);
            if (grow.first) { buckets = grow.second; out.push_back(Rehash{(uint64_t)count, (uint64_t)buckets}); }
        }
        return out;
    }
    // the closed form on the host: one sort per stretch between rehashes
    static std::vector<uint32_t> of_sorted(const uint64_t* keys, size_t n) {
        const std::vector<Rehash> sch = schedule(n);
        std::vector<uint32_t> list, seq, first;
        std::vector<uint64_t> sortkey;
        for (size_t j = 0; j < sch.size(); j++) {
            const size_t c = (size_t)sch[j].count, m = j + 1 < sch.size() ? (size_t)sch[j + 1].count : n, B = (size_t)sch[j].buckets;
            if (m == 0) continue;                                        // (the empty container's first rehash: nothing to relink)
            seq.assign(list.begin(), list.end());                        // the list as it stands (c nodes) ...
            for (size_t t = c; t < m; t++) seq.push_back((uint32_t)t);   // ... followed by the nodes inserted until the next rehash
            first.assign(B, 0xFFFFFFFFu);
            for (size_t t = m; t-- > 0;) first[keys[seq[t]] % B] = (uint32_t)t;
            sortkey.resize(m);
            for (size_t t = 0; t < m; t++) sortkey[t] = ((uint64_t)(0xFFFFFFFFu - first[keys[seq[t]] % B]) << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)t);
            std::sort(sortkey.begin(), sortkey.end());
            list.resize(m);
            for (size_t i = 0; i < m; i++) list[i] = seq[0xFFFFFFFFu - (uint32_t)sortkey[i]];
        }
        return list;
    }

private:
    enum : uint32_t {              // (enumerators, not static members: C++11 wants a definition for a static member that is bound to a reference)
        kNil = 0xFFFFFFFFu,        // no node
        kHead = 0xFFFFFFFEu        // the list's before-begin sentinel in the role of a bucket's "before" node
    };
    DumpOrder(const uint64_t* keys, size_t n) : keys_(keys), next_(n, (uint32_t)kNil), bucket_(1, (uint32_t)kNil) {}
    uint32_t& link_of(uint32_t before) { return before == kHead ? head_ : next_[before]; }
    void insert(uint32_t i) {
        const std::pair<bool, size_t> grow = policy_._M_need_rehash(bucket_.size(), count_, 1);
        if (grow.first) rehash(grow.second);
        const size_t b = keys_[i] % bucket_.size();
        if (bucket_[b] != kNil) {
            uint32_t& link = link_of(bucket_[b]);
            next_[i] = link;
            link = i;
        } else {
            next_[i] = head_;
            head_ = i;
            if (next_[i] != kNil) bucket_[keys_[next_[i]] % bucket_.size()] = i;
            bucket_[b] = kHead;
        }
        count_++;
    }
    void rehash(size_t n_buckets) {
        std::vector<uint32_t> fresh(n_buckets, (uint32_t)kNil);
        uint32_t p = head_;
        head_ = kNil;
        size_t first_bucket = 0;
        while (p != kNil) {
            const uint32_t following = next_[p];
            const size_t b = keys_[p] % n_buckets;
            if (fresh[b] == kNil) {
                next_[p] = head_;
                head_ = p;
                fresh[b] = kHead;
                if (next_[p] != kNil) fresh[first_bucket] = p;
                first_bucket = b;
            } else {
                uint32_t& link = fresh[b] == kHead ? head_ : next_[fresh[b]];
                next_[p] = link;
                link = p;
            }
            p = following;
        }
        bucket_.swap(fresh);
    }
    const uint64_t* keys_;
    std::vector<uint32_t> next_, bucket_;
    uint32_t head_ = kNil;
    size_t count_ = 0;
    std::__detail::_Prime_rehash_policy policy_;
};
