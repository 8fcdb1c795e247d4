// CPU check of faucet_amd/host/junction_order.h: the replayed dump order equals the iteration order of a real std::unordered_map filled with
// the same keys in the same order (the reference's container, utils/JunctionMap.h:61), across many rehashes and for crowded small key spaces.
// Built and run by tests/test_abi_cpu.py.
#include <stdio.h>

#include "junction_order.h"

int main() {
    uint64_t s = 88172645463325252ULL;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    const size_t sizes[] = {0, 1, 2, 11, 12, 13, 14, 28, 29, 30, 100, 1000, 54321, 400000};
    for (size_t n : sizes) {
        std::vector<uint64_t> keys(n);
        for (uint64_t& k : keys) k = rnd() >> 2;          // 62-bit k-mers (k = 31)
        if (!DumpOrder::agrees_with_the_container(keys.data(), n)) { printf("mismatch at n = %zu\n", n); return 1; }
    }
    for (size_t stride : {389, 7919, 1}) {                // k = 5: every key below 1024, buckets crowded whatever their number
        std::vector<uint64_t> keys;
        for (size_t i = 0; i < 1024 && keys.size() < 700; i++) keys.push_back((i * stride) % 1024);
        if (stride == 1) keys.resize(1024), keys[1023] = 1023;
        if (stride == 1) for (size_t i = 0; i < 1024; i++) keys[i] = 1023 - i;
        if (!DumpOrder::agrees_with_the_container(keys.data(), keys.size())) { printf("mismatch with small keys, stride %zu\n", stride); return 1; }
    }
    std::vector<uint64_t> twice = {5, 9, 5};              // repeated keys are not what the replay is for: it must say so
    if (DumpOrder::agrees_with_the_container(twice.data(), twice.size())) { printf("repeated keys accepted\n"); return 1; }
    printf("ok\n");
    return 0;
}
