/*
 * faucet_oracle.cpp — CPU restatement of Faucet's Bloom load pass and ReadScanner junction scan.
 *
 * TEST INFRASTRUCTURE ONLY (see faucet_oracle.h).  Written from the semantics of the reference
 * (SURVEY.md Appendix A), not from its text; each block cites the reference lines it follows.
 * Single-threaded on purpose: the reference has no threads, so this is also the "port" CPU
 * baseline that bench.py times next to the GPU path.
 */
#include "faucet_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <list>
#include <set>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

/* ------------------------------------------------------------------ codec */
/* A0 C1 T2 G3 via bits 1..2 of the ASCII code (utils/Kmer.cpp:82-88). */
inline int nt2int(char c) { return (static_cast<int>(c) >> 1) & 3; }
inline bool valid_nuc(char c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; } /* Kmer.cpp:50-60 */
inline uint64_t kmask(int k) { return k >= 32 ? ~0ULL : ((1ULL << (2 * k)) - 1); }     /* Kmer.cpp:37-48 */

/* Reverse complement: complement every 2-bit group (x ^ 2 per group = XOR 0xAA..), reverse the
 * groups, right-align (utils/Kmer.cpp:238-252 does it with a byte LUT; same function). */
inline uint64_t revcomp(uint64_t x, int k) {
    uint64_t y = x ^ 0xAAAAAAAAAAAAAAAAULL;
    y = ((y >> 2) & 0x3333333333333333ULL) | ((y & 0x3333333333333333ULL) << 2);
    y = ((y >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((y & 0x0F0F0F0F0F0F0F0FULL) << 4);
    y = __builtin_bswap64(y);
    return y >> (2 * (32 - k));
}
inline uint64_t canon(uint64_t x, int k) { return std::min(x, revcomp(x, k)); }       /* Kmer.cpp:531-533 */
inline uint64_t encode(const char* s, int k) {                                       /* Kmer.cpp:429-433 */
    uint64_t x = 0, m = kmask(k);
    for (int i = 0; i < k; i++) x = ((x << 2) + nt2int(s[i])) & m;
    return x;
}

/* ------------------------------------------------------------------ hash */
const uint64_t kRbase[10] = {                                                        /* Bloom.h:56-68 */
    0xAAAAAAAA55555555ULL, 0x33333333CCCCCCCCULL, 0x6666666699999999ULL, 0xB5B5B5B54B4B4B4BULL,
    0xAA55AA5555335533ULL, 0x33CC33CCCC66CC66ULL, 0x6699669999B599B5ULL, 0xB54BB54B4BAA4BAAULL,
    0xAA33AA3355CC55CCULL, 0x33663366CC99CC99ULL};

struct Seeds {
    uint64_t tab[10];
    Seeds() {                                                                        /* Bloom.cpp:500-511, user_seed = 0 */
        for (int i = 0; i < 10; i++) tab[i] = kRbase[i];
        for (int i = 0; i < 10; i++) tab[i] = tab[i] * tab[(i + 3) % 10] + 0;
    }
};
const Seeds kSeeds;

inline uint64_t old_hash(uint64_t key, int num, uint64_t mask) {                     /* Bloom.h:134-145 */
    uint64_t h = kSeeds.tab[num];
    h ^= (h << 7) ^ (key * (h >> 3)) ^ (~((h << 11) + (key ^ (h >> 5))));
    h = (~h) + (h << 21);
    h ^= h >> 24;
    h = (h + (h << 3)) + (h << 8);
    h ^= h >> 14;
    h = (h + (h << 2)) + (h << 4);
    h ^= h >> 28;
    h += h << 31;
    return h & mask;
}

}  // namespace

/* ------------------------------------------------------------------ Bloom */
struct fo_bloom {
    uint64_t tai = 0;
    int n_hash = 0;
    std::vector<uint8_t> bits;
    bool fake = false;
    std::set<uint64_t> fake_set;
    mutable uint64_t n_tests = 0, n_sets = 0;

    /* Bloom.h:217-226 */
    void add(uint64_t h0, uint64_t h1) {
        uint64_t h = h0;
        for (int i = 0; i < n_hash; i++, h += h1) {
            h %= tai;
            bits[h >> 3] |= static_cast<uint8_t>(1u << (h & 7));
            n_sets++;
        }
    }
    /* Bloom.h:242-258 (real filters; the fake-filter branch of :243-246 is never reached on the
     * paths restated here because fake filters are only queried through old_contains) */
    int contains(uint64_t h0, uint64_t h1) const {
        uint64_t h = h0 % tai;
        for (int i = 0; i < n_hash; i++, h = (h + h1) % tai) {
            n_tests++;
            if (!(bits[h >> 3] & (1u << (h & 7)))) return 0;
        }
        return 1;
    }
    int old_contains(uint64_t elem) const {                                          /* Bloom.h:162-173 */
        if (fake) return fake_set.count(elem) ? 1 : 0;
        return contains(old_hash(elem, 0, tai - 1), old_hash(elem, 1, tai - 1));
    }
    void old_add(uint64_t elem) { add(old_hash(elem, 0, tai - 1), old_hash(elem, 1, tai - 1)); }
};

namespace {

/* getUnambiguousReads (utils/Kmer.cpp:64-80): maximal ACGT runs of length >= k, LAST FIRST. */
struct Seg { uint64_t start, len; };
inline void unambiguous_segments(const char* s, uint64_t len, int k, std::vector<Seg>& out) {
    out.clear();
    uint64_t i = 0;
    while (i < len) {
        while (i < len && !valid_nuc(s[i])) i++;
        uint64_t b = i;
        while (i < len && valid_nuc(s[i])) i++;
        if (i - b >= static_cast<uint64_t>(k)) out.push_back({b, i - b});
    }
    std::reverse(out.begin(), out.end());   /* push_front at Kmer.cpp:77 */
}

/* Cursor over one ACGT string in half-steps: the (pos, direction) walker of utils/ReadKmer.cpp.
 * t = 2*pos + (facing forward ? 1 : 0)  (ReadKmer.cpp:33-35). */
struct Cursor {
    const char* s;
    int len, k;
    uint64_t mask;
    uint64_t fwd, rc;   /* DoubleKmer: forward strand and its reverse complement, in lock-step */
    int pos;
    bool forward_facing;

    Cursor(const char* str, int n, int kk) : s(str), len(n), k(kk), mask(kmask(kk)), pos(0), forward_facing(false) {
        fwd = encode(s, k);                                                          /* ReadKmer.cpp:121-128 */
        rc = revcomp(fwd, k);
    }
    Cursor(const char* str, int n, int kk, int index, bool dir) : s(str), len(n), k(kk), mask(kmask(kk)), pos(index), forward_facing(dir) {
        fwd = encode(s + index, k);                                                  /* ReadKmer.cpp:131-138 */
        rc = revcomp(fwd, k);
    }
    int t() const { return 2 * pos + (forward_facing ? 1 : 0); }
    int dist_to_end() const { return 2 * len - t() - 2 * k + 1; }                     /* ReadKmer.cpp:28-30 */
    void step() {                                                                    /* ReadKmer.cpp:72-83 */
        forward_facing = !forward_facing;
        if (forward_facing) return;
        int nuc = 0;
        if (pos + k < len) nuc = nt2int(s[pos + k]);
        fwd = ((fwd << 2) + static_cast<uint64_t>(nuc)) & mask;                       /* DoubleKmer.cpp:5-8 */
        int cn = nuc < 2 ? nuc + 2 : nuc - 2;                                        /* Kmer.cpp:90-93 */
        rc = ((rc >> 2) + (static_cast<uint64_t>(cn) << (2 * k - 2))) & mask;
        pos++;
    }
    uint64_t key() const { return forward_facing ? fwd : rc; }                       /* ReadKmer.cpp:50-57 */
    uint64_t canonical() const { return std::min(fwd, rc); }                         /* DoubleKmer.cpp:19-21 */
    int real_ext_nuc() const {                                                       /* ReadKmer.cpp:107-114 */
        if (forward_facing) return nt2int(pos + k < len ? s[pos + k] : '\0');
        int n = nt2int(s[pos - 1]);
        return n < 2 ? n + 2 : n - 2;
    }
    uint64_t ext(int nuc) const { return ((key() << 2) | static_cast<uint64_t>(nuc)) & mask; }   /* DoubleKmer.cpp:10-17 */
    uint64_t real_ext() const { return ext(real_ext_nuc()); }
    int ext_index(bool dir) const { return dir != forward_facing ? 4 : real_ext_nuc(); }          /* ReadKmer.cpp:95-100 */
};

struct Junction {   /* utils/Junction.h:10-18, Junction.cpp:59-72 */
    uint8_t cov[4] = {0, 0, 0, 0};
    uint8_t dist[5] = {0, 0, 0, 0, 0};
    bool linked[5] = {false, false, false, false, false};
    void add_coverage(int e) {
        cov[e] = static_cast<uint8_t>(cov[e] + 1);
        if (cov[e] == 0) cov[e] = 255;
    }
    void update(int e, int length) {   /* the int argument narrows to unsigned char at the call (Junction.cpp:69) */
        uint8_t l = static_cast<uint8_t>(length);
        dist[e] = std::max(dist[e], l);
    }
};

}  // namespace

struct fo_scanner {
    int k, j, max_spacer;
    fo_bloom* bloom;
    fo_bloom* short_pf;
    fo_bloom* long_pf;
    std::unordered_map<uint64_t, Junction> map;      /* utils/JunctionMap.h:61 */
    std::vector<uint64_t> creation_order;
    fo_scan_stats st;
    bool first_end = true;
    std::list<uint64_t> back1, back2;
    uint64_t bit_tests_valid = 0;   /* bit tests spent inside getValidReads (the rest of bloom->n_tests is junction probing) */
    /* optional log, one entry per element of scanInputRead's list: ReadKmer::pos inside the valid piece | facing FORWARD << 28 | first
     * element of its scan_forward call << 29 | add_fake_junction's element << 30 (the layout of fgpu_stop::info, include/faucet_gpu.h) */
    std::vector<uint32_t>* visit_log = nullptr;

    Junction* find(uint64_t key) {
        auto it = map.find(key);
        return it == map.end() ? nullptr : &it->second;
    }
    Junction* create(uint64_t key) {                                                 /* JunctionMap.cpp:567-570 */
        auto r = map.insert(std::pair<uint64_t, Junction>(key, Junction()));
        if (r.second) creation_order.push_back(key);
        return &r.first->second;
    }

    /* JChecker::jcheck(kmer_type) (utils/JChecker.cpp:51-80): is there a chain of j forward
     * extensions that are all in the filter?  Level-by-level as in the reference. */
    bool jcheck(uint64_t kmer) {
        std::vector<uint64_t> last{kmer}, next;
        uint64_t mask = kmask(k);
        for (int lvl = 0; lvl < j; lvl++) {
            next.clear();
            for (uint64_t km : last)
                for (int nt = 0; nt < 4; nt++) {
                    uint64_t e = ((km << 2) + static_cast<uint64_t>(nt)) & mask;
                    if (bloom->old_contains(canon(e, k))) next.push_back(e);
                }
            if (next.empty()) return false;
            last.swap(next);
        }
        return true;
    }

    /* testForJunction (src/ReadScanner.cpp:36-56) */
    bool test_for_junction(const Cursor& c, int* njcheck = nullptr) {
        uint64_t real = c.real_ext();
        for (int nt = 0; nt < 4; nt++) {
            uint64_t e = c.ext(nt);
            if (e == real) continue;
            if (bloom->old_contains(canon(e, k))) {
                st.nb_jcheck_kmer++;
                if (njcheck) (*njcheck)++;
                if (jcheck(e)) return true;
            }
        }
        return false;
    }

    /* find_next_junction (:61-86) */
    bool find_next_junction(Cursor& c, int last_junc_pos) {
        for (; c.dist_to_end() > 2 * j; c.step()) {
            if (map.find(c.key()) != map.end()) return true;
            if (c.t() - last_junc_pos >= 2 * max_spacer - 1) return true;
            if (test_for_junction(c)) return true;
            st.nb_processed++;
        }
        return false;
    }

    /* add_fake_junction (:92-104) */
    uint64_t add_fake_junction(const char* s, int len) {
        Cursor m(s, len, k, len / 2 - k / 2, true);
        uint64_t extension = m.real_ext();
        Junction* jn = create(m.key());
        jn->add_coverage(m.real_ext_nuc());
        jn->update(m.ext_index(false), m.t() - 2 * j);
        jn->update(m.ext_index(true), m.dist_to_end() - 2 * j);
        return extension;
    }

    /* scan_forward (:112-231) on one valid piece */
    std::list<uint64_t> scan_forward(const char* s, int len, bool no_cleaning) {
        std::list<uint64_t> result;
        Cursor c(s, len, k);
        for (int i = 0; i < 2 * j + 1; i++) c.step();

        bool have_last = false, have_first_back = false, have_last_fwd = false;
        Cursor last_k = c, first_back = c, last_fwd = c;
        uint64_t last_key = 0;
        int rev_pos = 0, for_pos = 0, last_junc_pos = 0;

        while (find_next_junction(c, last_junc_pos)) {
            Junction* junc = find(c.key());
            last_junc_pos = c.t();
            if (!junc) junc = create(c.key());
            if (visit_log) visit_log->push_back(static_cast<uint32_t>(c.pos) | (c.forward_facing ? 1u << 28 : 0u) | (result.empty() ? 1u << 29 : 0u));
            result.push_back(c.real_ext());

            if (!c.forward_facing) {                                                 /* :147-157 */
                if (!have_first_back) { have_first_back = true; first_back = c; rev_pos = c.pos; }
            } else {                                                                 /* :158-169 */
                if (!have_last_fwd) { have_last_fwd = true; for_pos = c.pos; }
                last_fwd = c;
            }
            junc->add_coverage(c.real_ext_nuc());                                    /* :171 */

            if (have_last) {                                                         /* directLinkJunctions, JunctionMap.cpp:551-561 */
                Junction* lj = find(last_key);
                int e1 = last_k.ext_index(true), e2 = c.ext_index(false);
                int d = c.t() - last_k.t();
                lj->update(e1, d);
                junc->update(e2, d);
                lj->linked[e1] = true;
                junc->linked[e2] = true;
            } else {                                                                 /* :178-183 */
                have_last = true;
                junc->update(c.ext_index(false), c.t() - 2 * j);
            }
            last_k = c;
            last_key = c.key();

            int idx = c.ext_index(true);                                             /* :188-192 */
            int d = std::max(1, static_cast<int>(junc->dist[idx]));
            for (int i = 0; i < d; i++) c.step();
            st.nb_processed++;
            st.nb_skipped += static_cast<uint64_t>(d - 1);
        }

        if (!have_last) {                                                            /* :195-200 */
            st.nb_no_juncs++;
            if (visit_log) visit_log->push_back(static_cast<uint32_t>(len / 2 - k / 2) | (1u << 28) | (1u << 29) | (1u << 30));
            result.push_back(add_fake_junction(s, len));
        } else {                                                                     /* :202-206 */
            find(last_key)->update(last_k.ext_index(true), last_k.dist_to_end() - 2 * j);
        }
        if (!no_cleaning) {                                                          /* :208-225 */
            if (result.size() == 2) {
                if (have_first_back && have_last_fwd && !(rev_pos > for_pos))
                    fo_bloom_add_pair(short_pf, first_back.real_ext(), last_fwd.real_ext(), k);
                if ((have_first_back && !have_last_fwd) || (!have_first_back && have_last_fwd))
                    fo_bloom_add_pair(short_pf, result.front(), result.back(), k);
            } else if (result.size() > 2) {
                std::vector<uint64_t> v(result.begin(), result.end());
                for (size_t i = 0; i + 2 < v.size(); i++) fo_bloom_add_pair(short_pf, v[i], v[i + 2], k);
            }
        }
        return result;
    }

    /* getValidReads (:233-257): runs of >= k consecutive windows present in the filter */
    void valid_pieces(const char* s, int len, std::vector<Seg>& out) {
        out.clear();
        int start = 0, end = 0;
        Cursor c(s, len, k);
        for (; c.dist_to_end() >= 0; c.step(), c.step()) {
            if (bloom->old_contains(c.canonical())) {
                end++;
            } else {
                if (end >= start + k) out.push_back({static_cast<uint64_t>(start), static_cast<uint64_t>(end - start + k - 1)});
                start = c.pos + 1;
                end = c.pos + 1;
            }
        }
        if (end >= start + k) out.push_back({static_cast<uint64_t>(start), static_cast<uint64_t>(end - start + k - 1)});
    }

    /* scanInputRead (:260-282) */
    std::list<uint64_t> scan_input_read(const char* line, uint64_t len, bool no_cleaning) {
        std::list<uint64_t> result;
        std::vector<Seg> segs, pieces;
        unambiguous_segments(line, len, k, segs);
        for (const Seg& sg : segs) {
            if (sg.len >= static_cast<uint64_t>(k + 2 * j + 1)) {
                st.unambiguous_reads++;
                const uint64_t tests_before = bloom->n_tests;
                valid_pieces(line + sg.start, static_cast<int>(sg.len), pieces);
                bit_tests_valid += bloom->n_tests - tests_before;
                for (const Seg& p : pieces) {
                    std::list<uint64_t> r = scan_forward(line + sg.start + p.start, static_cast<int>(p.len), no_cleaning);
                    result.splice(result.end(), r);
                    st.reads_no_errors++;
                }
            }
        }
        return result;
    }

    /* body of the loop in scanReads (:304-351) for one record */
    void scan_record(const char* line, uint64_t len, bool paired_ends, bool no_cleaning) {
        if (first_end) back1 = scan_input_read(line, len, no_cleaning);
        else back2 = scan_input_read(line, len, no_cleaning);
        if (paired_ends && !first_end) {
            if (!back1.empty() && !back2.empty()) {
                st.not_empty_count++;
                for (uint64_t pair1 : back1) {
                    bool paired = false;
                    if (!no_cleaning) {
                        for (uint64_t pair2 : back2)
                            if (fo_bloom_contains_pair(long_pf, pair1, pair2, k)) { paired = true; break; }
                        if (!paired) fo_bloom_add_pair(long_pf, pair1, back2.front(), k);
                    }
                }
            } else {
                st.empty_count++;
            }
        }
        st.reads_processed++;
        first_end = !first_end;
    }
};

/* ======================================================================= C interface */
extern "C" {

int fo_nt2int(char c) { return nt2int(c); }
int fo_is_valid_nuc(char c) { return valid_nuc(c) ? 1 : 0; }
uint64_t fo_encode(const char* s, int k) { return encode(s, k); }
uint64_t fo_revcomp(uint64_t x, int k) { return revcomp(x, k); }
uint64_t fo_canon(uint64_t x, int k) { return canon(x, k); }
void fo_decode(uint64_t x, int k, char* out) {
    static const char tab[4] = {'A', 'C', 'T', 'G'};                                 /* Kmer.cpp:21 */
    for (int i = k - 1; i >= 0; i--) { out[i] = tab[x & 3]; x >>= 2; }
    out[k] = 0;
}

uint64_t fo_seed(int i) { return kSeeds.tab[i]; }
uint64_t fo_old_hash(uint64_t key, int num, uint64_t tai) { return old_hash(key, num, tai - 1); }

uint64_t fo_bloom_tai(uint64_t requested) {                                          /* Bloom.cpp:173-178 */
    int hash_size = static_cast<int>(log2(static_cast<double>(requested))) + 1;
    uint64_t tai = static_cast<uint64_t>(pow(2, hash_size));
    if (tai == 0) tai = 1;
    return tai;
}

/* Brent root finder with the reference's exact control flow (Bloom.cpp:33-124): returns the
 * LAST iterate s at the moment |b-a| < tol, not the bracket end. */
static double brent(double (*f)(double, const void*), const void* ctx, double lower, double upper, double tol,
                    unsigned max_iter, int* iters) {
    double a = lower, b = upper;
    double fa = f(a, ctx), fb = f(b, ctx), fs = 0;
    if (!(fa * fb < 0)) return -11;
    if (std::abs(fa) < std::abs(b)) { std::swap(a, b); std::swap(fa, fb); }          /* sic: |b|, Bloom.cpp:47 */
    double c = a, fc = fa, s = 0, d = 0;
    bool mflag = true;
    for (unsigned iter = 1; iter < max_iter; ++iter) {
        if (std::abs(b - a) < tol) { if (iters) *iters = static_cast<int>(iter); return s; }
        if (fa != fc && fb != fc)
            s = (a * fb * fc / ((fa - fb) * (fa - fc))) + (b * fa * fc / ((fb - fa) * (fb - fc))) +
                (c * fa * fb / ((fc - fa) * (fc - fb)));
        else
            s = b - fb * (b - a) / (fb - fa);
        if (((s < (3 * a + b) * 0.25) || (s > b)) || (mflag && (std::abs(s - b) >= (std::abs(b - c) * 0.5))) ||
            (!mflag && (std::abs(s - b) >= (std::abs(c - d) * 0.5))) || (mflag && (std::abs(b - c) < tol)) ||
            (!mflag && (std::abs(c - d) < tol))) {
            s = (a + b) * 0.5;
            mflag = true;
        } else {
            mflag = false;
        }
        fs = f(s, ctx);
        d = c;
        c = b;
        fc = fb;
        if (fa * fs < 0) { b = s; fb = fs; } else { a = s; fa = fs; }
        if (std::abs(fa) < std::abs(fb)) { std::swap(a, b); std::swap(fa, fb); }
    }
    if (iters) *iters = -1;
    return -12;   /* the reference falls off the end here (UB); callers treat < 0 as failure */
}

struct P1Ctx { uint64_t e, s; float fp; };
static double p1_func(double p1, const void* vctx) {                                 /* src/Faucet.cpp:197-201 */
    const P1Ctx* x = static_cast<const P1Ctx*>(vctx);
    double c = (x->e - (1 - p1) * x->s) / x->e;
    return log(2) * std::log(x->fp) + log(p1) * log(1 - pow(2, -c));                 /* std::log(float) -> float, as there */
}
double fo_solve_p1(uint64_t estimated, uint64_t singletons, float fp, int* iterations) {
    P1Ctx ctx{estimated, singletons, fp};
    return brent(p1_func, &ctx, fp, 0.50, 0.0001, 1000, iterations);                  /* Faucet.cpp:209 */
}

void fo_size_optimal(uint64_t estimated, float fp, int* bits_per_item, uint64_t* tai, int* n_hash) {
    int bits = -std::log(fp) / log(2) / log(2);                                      /* Bloom.cpp:232 (float log, double division) */
    uint64_t size = static_cast<uint64_t>(estimated * bits);                         /* :236 */
    int nh = static_cast<int>(floorf(0.7 * bits));                                   /* :243-244 */
    if (nh > 10 || nh < 1) nh = 4;                                                   /* :491-498 keeps the ctor default (:170) */
    if (bits_per_item) *bits_per_item = bits;
    if (tai) *tai = fo_bloom_tai(size);
    if (n_hash) *n_hash = nh;
}
void fo_size_two_hash(uint64_t estimated, float fp, int* bits_per_item, uint64_t* tai, int* n_hash) {
    int bits = 2 * static_cast<int>(1 / pow(fp, .5));                                /* Bloom.cpp:209 */
    uint64_t size = static_cast<uint64_t>(estimated * bits);
    if (bits_per_item) *bits_per_item = bits;
    if (tai) *tai = fo_bloom_tai(size);
    if (n_hash) *n_hash = 2;
}

fo_bloom* fo_bloom_new(uint64_t tai, int n_hash) {
    fo_bloom* b = new fo_bloom();
    b->tai = tai;
    b->n_hash = n_hash;
    b->bits.assign(tai / 8 ? tai / 8 : 1, 0);
    return b;
}
void fo_bloom_free(fo_bloom* b) { delete b; }
uint8_t* fo_bloom_bits(fo_bloom* b) { return b->bits.data(); }
uint64_t fo_bloom_nbytes(const fo_bloom* b) { return b->tai / 8; }
float fo_bloom_weight(const fo_bloom* b) {                                           /* Bloom.cpp:191-203 */
    long w = 0;
    for (uint64_t i = 0; i < b->tai / 8; i++) w += __builtin_popcount(b->bits[i]);
    return static_cast<float>(w) / static_cast<float>(b->tai);
}
void fo_bloom_fakify(fo_bloom* b) { b->fake = true; }
void fo_bloom_add_fake(fo_bloom* b, uint64_t c) { b->fake_set.insert(c); }
void fo_bloom_old_add(fo_bloom* b, uint64_t c) { b->old_add(c); }
int fo_bloom_old_contains(fo_bloom* b, uint64_t c) { return b->old_contains(c); }
void fo_bloom_add_pair(fo_bloom* b, uint64_t k1, uint64_t k2, int k) {               /* Bloom.cpp:127-139 */
    uint64_t e1 = canon(k1, k), e2 = canon(k2, k);
    b->add(old_hash(std::min(e1, e2), 0, b->tai - 1), old_hash(std::max(e1, e2), 1, b->tai - 1));
}
int fo_bloom_contains_pair(fo_bloom* b, uint64_t k1, uint64_t k2, int k) {           /* Bloom.cpp:141-154 */
    uint64_t e1 = canon(k1, k), e2 = canon(k2, k);
    return b->contains(old_hash(std::min(e1, e2), 0, b->tai - 1), old_hash(std::max(e1, e2), 1, b->tai - 1));
}
uint64_t fo_bloom_bit_tests(const fo_bloom* b) { return b->n_tests; }
uint64_t fo_bloom_bit_sets(const fo_bloom* b) { return b->n_sets; }
void fo_bloom_reset_counters(fo_bloom* b) { b->n_tests = b->n_sets = 0; }

int fo_reads_from_file(const char* path, int fastq, fo_reads* out) {
    std::ifstream in(path);
    if (!in.is_open()) return -1;
    std::string line, all;
    std::vector<uint64_t> offs{0};
    while (std::getline(in, line)) {                /* header (any line) */
        std::getline(in, line);                     /* sequence; empty if the stream just ended */
        all += line;
        offs.push_back(all.size());
        if (fastq) { std::getline(in, line); std::getline(in, line); }
        line.clear();
    }
    out->n = offs.size() - 1;
    out->bases = static_cast<char*>(malloc(all.size() ? all.size() : 1));
    memcpy(out->bases, all.data(), all.size());
    out->offsets = static_cast<uint64_t*>(malloc(offs.size() * sizeof(uint64_t)));
    memcpy(out->offsets, offs.data(), offs.size() * sizeof(uint64_t));
    return 0;
}
void fo_reads_free(fo_reads* r) {
    free(r->bases);
    free(r->offsets);
    r->bases = nullptr;
    r->offsets = nullptr;
    r->n = 0;
}

void fo_load_two_filters(fo_bloom* bloo1, fo_bloom* bloo2, const char* bases, const uint64_t* offsets, uint64_t n,
                         int k, fo_load_stats* stats) {
    fo_load_stats st{0, 0, 0, 0};
    std::vector<Seg> segs;
    const uint64_t mask = bloo1->tai - 1;
    for (uint64_t r = 0; r < n; r++) {
        const char* line = bases + offsets[r];
        unambiguous_segments(line, offsets[r + 1] - offsets[r], k, segs);
        for (const Seg& sg : segs) {
            st.unambiguous_reads++;
            Cursor c(line + sg.start, static_cast<int>(sg.len), k);
            for (; c.dist_to_end() >= 0; c.step(), c.step()) {                       /* Bloom.cpp:289-298 */
                uint64_t cn = c.canonical();
                uint64_t ha = old_hash(cn, 0, mask), hb = old_hash(cn, 1, mask);
                st.kmers++;
                if (bloo1->contains(ha, hb)) { bloo2->add(ha, hb); st.to_bloo2++; }
                else bloo1->add(ha, hb);
            }
        }
        st.reads_processed++;
    }
    if (stats) *stats = st;
}

/* load_two_filters with mercy == true (utils/Bloom.cpp:300-333): low-coverage k-mers between two solid ones are added to bloo2
 * as well unless the solid k-mer next to them looks like a junction in bloo1 (isJunction, :249-265).  Restated with its two
 * peculiarities: `last_kmer` points at the loop's own cursor (so it always IS the current k-mer), and the cursor faces
 * BACKWARD whenever the loop body runs (ReadKmer starts facing backward and is advanced two half-steps per iteration), so
 * getRealExtension() is the real BACKWARD extension in both calls, whatever `dir` says. */
static bool mercy_is_junction(const Cursor& c, fo_bloom* bloom, bool dir_forward) {
    const uint64_t real_ext = c.real_ext();                        /* cursor faces backward: reverse complement of the window before */
    for (int nt = 0; nt < 4; nt++) {
        const uint64_t test_ext = (((dir_forward ? c.fwd : c.rc) << 2) | static_cast<uint64_t>(nt)) & c.mask;   /* DoubleKmer.cpp:10-17 */
        if (real_ext != test_ext && bloom->old_contains(canon(test_ext, c.k))) return true;
    }
    return false;
}

void fo_load_two_filters_mercy(fo_bloom* bloo1, fo_bloom* bloo2, const char* bases, const uint64_t* offsets, uint64_t n,
                               int k, fo_load_stats* stats) {
    fo_load_stats st{0, 0, 0, 0};
    std::vector<Seg> segs;
    const uint64_t mask = bloo1->tai - 1;
    for (uint64_t r = 0; r < n; r++) {
        const char* line = bases + offsets[r];
        unambiguous_segments(line, offsets[r + 1] - offsets[r], k, segs);
        for (const Seg& sg : segs) {
            st.unambiguous_reads++;
            bool have_last = false;
            std::vector<std::pair<uint64_t, uint64_t>> hash_vals;
            Cursor c(line + sg.start, static_cast<int>(sg.len), k);
            for (; c.dist_to_end() >= 0; c.step(), c.step()) {                       /* Bloom.cpp:303-331 */
                uint64_t cn = c.canonical();
                uint64_t ha = old_hash(cn, 0, mask), hb = old_hash(cn, 1, mask);
                st.kmers++;
                if (bloo1->contains(ha, hb)) {
                    bloo2->add(ha, hb);
                    st.to_bloo2++;
                    have_last = true;
                    if (!hash_vals.empty()) {                                        /* came from low to high */
                        if (!mercy_is_junction(c, bloo1, false))
                            for (const auto& v : hash_vals) bloo2->add(v.first, v.second);
                        hash_vals.clear();
                    }
                } else {
                    bloo1->add(ha, hb);
                    if (have_last) {
                        if (hash_vals.empty()) {                                     /* came from high to low */
                            if (!mercy_is_junction(c, bloo1, true)) hash_vals.emplace_back(ha, hb);
                        } else {
                            hash_vals.emplace_back(ha, hb);
                        }
                    }
                }
            }
        }
        st.reads_processed++;
    }
    if (stats) *stats = st;
}

/* ---- Stage 3's Bloom probes (pure functions of the filter) ---- */
static bool stage3_jcheck(fo_bloom* b, uint64_t kmer, int k, int j) {                /* JChecker::jcheck(kmer_type), JChecker.cpp:51-80 */
    const uint64_t mask = kmask(k);
    std::vector<uint64_t> last{kmer}, next;
    for (int lvl = 0; lvl < j; lvl++) {
        next.clear();
        for (uint64_t km : last)
            for (int nt = 0; nt < 4; nt++) {
                uint64_t e = ((km << 2) + static_cast<uint64_t>(nt)) & mask;
                if (b->old_contains(canon(e, k))) next.push_back(e);
            }
        if (next.empty()) return false;
        last.swap(next);
    }
    return true;
}
int fo_stage3_jcheck(fo_bloom* b, uint64_t kmer, int k, int j) { return stage3_jcheck(b, kmer, k, j) ? 1 : 0; }
int fo_stage3_valid_extension(fo_bloom* b, uint64_t kmer, int k, int j) {            /* JunctionMap::getValidJExtension, JunctionMap.cpp:474-490 */
    const uint64_t mask = kmask(k);
    int answer = -1;
    for (int i = 0; i < 4; i++) {
        uint64_t e = ((kmer << 2) + static_cast<uint64_t>(i)) & mask;
        if (b->old_contains(canon(e, k)) && stage3_jcheck(b, e, k, j)) {
            if (answer != -1) return -2;
            answer = i;
        }
    }
    return answer;
}
int fo_stage3_bloom_junction(fo_bloom* b, uint64_t kmer, int k, int j) {             /* JunctionMap::isBloomJunction, JunctionMap.cpp:494-504 */
    const uint64_t mask = kmask(k);
    int paths = 0;
    for (int i = 0; i < 4; i++)
        if (stage3_jcheck(b, ((kmer << 2) + static_cast<uint64_t>(i)) & mask, k, j)) paths++;
    return paths > 1 ? 1 : 0;
}

void fo_load_single_filter(fo_bloom* bloo1, const char* bases, const uint64_t* offsets, uint64_t n, int k,
                           fo_load_stats* stats) {
    fo_load_stats st{0, 0, 0, 0};
    std::vector<Seg> segs;
    for (uint64_t r = 0; r < n; r++) {
        const char* line = bases + offsets[r];
        unambiguous_segments(line, offsets[r + 1] - offsets[r], k, segs);
        for (const Seg& sg : segs) {
            st.unambiguous_reads++;
            Cursor c(line + sg.start, static_cast<int>(sg.len), k);
            for (; c.dist_to_end() >= 0; c.step(), c.step()) { bloo1->old_add(c.canonical()); st.kmers++; }
        }
        st.reads_processed++;
    }
    if (stats) *stats = st;
}

fo_scanner* fo_scanner_new(int k, int j, int max_spacer_dist, fo_bloom* bloom, fo_bloom* spf, fo_bloom* lpf) {
    fo_scanner* s = new fo_scanner();
    s->k = k;
    s->j = j;
    s->max_spacer = max_spacer_dist;
    s->bloom = bloom;
    s->short_pf = spf;
    s->long_pf = lpf;
    memset(&s->st, 0, sizeof(s->st));
    return s;
}
void fo_scanner_free(fo_scanner* s) { delete s; }

void fo_scan_reads(fo_scanner* s, const char* bases, const uint64_t* offsets, uint64_t n, int paired_ends, int no_cleaning) {
    for (uint64_t r = 0; r < n; r++)
        s->scan_record(bases + offsets[r], offsets[r + 1] - offsets[r], paired_ends != 0, no_cleaning != 0);
}
uint64_t fo_scan_input_read_ex(fo_scanner* s, const char* line, uint64_t len, int no_cleaning, uint64_t* ext_out, uint32_t* info_out, uint64_t cap) {
    std::vector<uint32_t> log;
    s->visit_log = &log;
    std::list<uint64_t> r = s->scan_input_read(line, len, no_cleaning != 0);
    s->visit_log = nullptr;
    uint64_t n = 0;
    for (uint64_t e : r) {
        if (n < cap) { ext_out[n] = e; info_out[n] = log[n]; }
        n++;
    }
    return n;
}

uint64_t fo_scan_input_read(fo_scanner* s, const char* line, uint64_t len, int no_cleaning, uint64_t* ext_out, uint64_t cap) {
    std::list<uint64_t> r = s->scan_input_read(line, len, no_cleaning != 0);
    uint64_t i = 0;
    for (uint64_t e : r) { if (i < cap && ext_out) ext_out[i] = e; i++; }
    return r.size();
}
void fo_scan_get_stats(const fo_scanner* s, fo_scan_stats* out) {
    *out = s->st;
    out->n_junctions = s->map.size();
}
static void fill_rec(const Junction& j, fo_junction* r) {
    for (int i = 0; i < 4; i++) r->cov[i] = j.cov[i];
    for (int i = 0; i < 5; i++) { r->dist[i] = j.dist[i]; r->linked[i] = j.linked[i] ? 1 : 0; }
}
uint64_t fo_scan_get_junctions(const fo_scanner* s, int order, uint64_t* keys, fo_junction* recs, uint64_t cap) {
    uint64_t i = 0;
    if (order == 0) {
        for (auto it = s->map.begin(); it != s->map.end() && i < cap; ++it, ++i) {
            if (keys) keys[i] = it->first;
            if (recs) fill_rec(it->second, &recs[i]);
        }
    } else {
        for (uint64_t key : s->creation_order) {
            if (i >= cap) break;
            if (keys) keys[i] = key;
            if (recs) fill_rec(s->map.at(key), &recs[i]);
            i++;
        }
    }
    return s->map.size();
}
uint64_t fo_scan_bit_tests_valid(const fo_scanner* s) { return s->bit_tests_valid; }
/* test-only: start a scanner from a junction map handed over by another process (records in creation order) */
void fo_scan_import(fo_scanner* s, const uint64_t* keys, const fo_junction* recs, uint64_t n, const fo_scan_stats* carried) {
    for (uint64_t i = 0; i < n; i++) {
        Junction* j = s->create(keys[i]);
        for (int c = 0; c < 4; c++) j->cov[c] = recs[i].cov[c];
        for (int c = 0; c < 5; c++) { j->dist[c] = recs[i].dist[c]; j->linked[c] = recs[i].linked[c] != 0; }
    }
    if (carried) s->st = *carried;
}
int fo_scan_write_junctions(const fo_scanner* s, const char* path) {                 /* JunctionMap.cpp:579-596, Junction.cpp:74-89 */
    FILE* f = fopen(path, "wb");
    if (!f) return -1;
    char buf[40];
    for (auto it = s->map.begin(); it != s->map.end(); ++it) {
        const Junction& j = it->second;
        fo_decode(it->first, s->k, buf);
        fprintf(f, "%s ", buf);
        for (int i = 0; i < 5; i++) fprintf(f, "%d ", j.dist[i]);
        fprintf(f, " ");
        for (int i = 0; i < 4; i++) fprintf(f, "%d ", j.cov[i]);
        fprintf(f, "%d ", j.cov[0] + j.cov[1] + j.cov[2] + j.cov[3]);                 /* getCoverage(4), Junction.cpp:48-53 */
        fprintf(f, " ");
        for (int i = 0; i < 5; i++) fprintf(f, "%d ", j.linked[i] ? 1 : 0);
        fprintf(f, "\n");
    }
    fclose(f);
    return 0;
}

uint64_t fo_get_valid_reads(fo_scanner* s, const char* seg, uint64_t len, uint64_t* out, uint64_t cap) {
    std::vector<Seg> p;
    s->valid_pieces(seg, static_cast<int>(len), p);
    for (uint64_t i = 0; i < p.size() && i < cap; i++) { out[2 * i] = p[i].start; out[2 * i + 1] = p[i].len; }
    return p.size();
}
int fo_test_for_junction(fo_scanner* s, const char* piece, uint64_t len, int t, int* njcheck) {
    Cursor c(piece, static_cast<int>(len), s->k);
    for (int i = 0; i < t; i++) c.step();
    uint64_t saved = s->st.nb_jcheck_kmer;
    int n = 0;
    bool f = s->test_for_junction(c, &n);
    s->st.nb_jcheck_kmer = saved;
    if (njcheck) *njcheck = n;
    return f ? 1 : 0;
}

}  // extern "C"
