// sizing.cpp — Bloom filter sizing, host arithmetic only (product code; the oracle has its own copy).
//
// Reproduces, truncation for truncation, what the reference computes once per run:
//   my_func / brents_fun call           src/Faucet.cpp:197-209, utils/Bloom.cpp:33-124
//   create_bloom_filter_optimal         utils/Bloom.cpp:229-247
//   create_bloom_filter_2_hash          utils/Bloom.cpp:206-226
//   Bloom::Bloom(tai_bloom, k)          utils/Bloom.cpp:165-181   (hashSize = (int)log2(size)+1: a power of two doubles)
//   set_number_of_hash_func             utils/Bloom.cpp:491-498   (values outside 1..10 keep the default 4)
// The false-positive rate is a float in the reference (src/Faucet.h:14) and its logarithm is taken in float.
#include <cmath>
#include <cstdint>
#include <utility>

#include "../../include/faucet_gpu.h"

namespace {

struct P1 {
    uint64_t e, s;
    float fp;
    double operator()(double p1) const {
        double c = (e - (1 - p1) * s) / e;
        return std::log(2.0) * static_cast<double>(std::log(fp)) + std::log(p1) * std::log(1 - std::pow(2.0, -c));
    }
};

// Brent's method with the reference's bookkeeping: the value returned is the last iterate computed before the
// bracket got narrower than tol, and the initial swap compares |f(a)| with |b| (sic).
template <class F>
double brent(const F& f, double lower, double upper, double tol, unsigned max_iter, int32_t* iterations) {
    double a = lower, b = upper, fa = f(a), fb = f(b), fs = 0;
    if (!(fa * fb < 0)) return -11;
    if (std::fabs(fa) < std::fabs(b)) { std::swap(a, b); std::swap(fa, fb); }
    double c = a, fc = fa, s = 0, d = 0;
    bool mflag = true;
    for (unsigned it = 1; it < max_iter; ++it) {
        if (std::fabs(b - a) < tol) {
            if (iterations) *iterations = static_cast<int32_t>(it);
            return s;
        }
        if (fa != fc && fb != fc)
            s = (a * fb * fc / ((fa - fb) * (fa - fc))) + (b * fa * fc / ((fb - fa) * (fb - fc))) + (c * fa * fb / ((fc - fa) * (fc - fb)));
        else
            s = b - fb * (b - a) / (fb - fa);
        const bool bisect = ((s < (3 * a + b) * 0.25) || (s > b)) || (mflag && (std::fabs(s - b) >= (std::fabs(b - c) * 0.5))) ||
                            (!mflag && (std::fabs(s - b) >= (std::fabs(c - d) * 0.5))) || (mflag && (std::fabs(b - c) < tol)) ||
                            (!mflag && (std::fabs(c - d) < tol));
        if (bisect) s = (a + b) * 0.5;
        mflag = bisect;
        fs = f(s);
        d = c;
        c = b;
        fc = fb;
        if (fa * fs < 0) { b = s; fb = fs; } else { a = s; fa = fs; }
        if (std::fabs(fa) < std::fabs(fb)) { std::swap(a, b); std::swap(fa, fb); }
    }
    if (iterations) *iterations = -1;
    return -12;
}

}  // namespace

extern "C" {

double fgpu_solve_p1(uint64_t estimated_kmers, uint64_t singletons, float fp, int32_t* iterations) {
    return brent(P1{estimated_kmers, singletons, fp}, fp, 0.50, 0.0001, 1000, iterations);
}

uint64_t fgpu_bloom_tai(uint64_t requested_bits) {
    int hash_size = static_cast<int>(std::log2(static_cast<double>(requested_bits))) + 1;
    uint64_t tai = static_cast<uint64_t>(std::pow(2.0, hash_size));
    return tai ? tai : 1;
}

void fgpu_size_optimal(uint64_t estimated, float fp, int32_t* bits_per_item, uint64_t* tai, int32_t* n_hash) {
    int bits = static_cast<int>(-static_cast<double>(std::log(fp)) / std::log(2.0) / std::log(2.0));
    int nh = static_cast<int>(floorf(static_cast<float>(0.7 * bits)));
    if (nh > 10 || nh < 1) nh = 4;
    if (bits_per_item) *bits_per_item = bits;
    if (tai) *tai = fgpu_bloom_tai(static_cast<uint64_t>(estimated * static_cast<uint64_t>(bits)));
    if (n_hash) *n_hash = nh;
}

void fgpu_size_two_hash(uint64_t estimated, float fp, int32_t* bits_per_item, uint64_t* tai, int32_t* n_hash) {
    int bits = 2 * static_cast<int>(1 / std::pow(static_cast<double>(fp), .5));
    if (bits_per_item) *bits_per_item = bits;
    if (tai) *tai = fgpu_bloom_tai(static_cast<uint64_t>(estimated * static_cast<uint64_t>(bits)));
    if (n_hash) *n_hash = 2;
}

}  // extern "C"
