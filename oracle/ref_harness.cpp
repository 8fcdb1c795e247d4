/*
 * ref_harness.cpp — OUR driver around the compiled reference objects (oracle/_ref/obj/*.o).
 *
 * TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile only when /root/reference is present; the
 * binary lands in oracle/_ref/ (git-ignored).  It calls the reference's own public functions and
 * prints known answers as JSON lines, which tests/golden/make_golden.py stores as fixtures:
 *   - codec/hash KATs          (utils/Kmer.cpp, utils/Bloom.h:134-145)
 *   - sizing KATs              (utils/Bloom.cpp:165-247, src/Faucet.cpp:197-209)
 *   - the ReadscanTest cases   (src/newTests/ReadscanTest.cpp:102-264) replayed through
 *     ReadScanner::scanInputRead with a fake Bloom, dumping the whole junction map
 *   - getValidReads / testForJunction probes on a real filter
 * No reference source text is copied here; this file only includes the reference headers.
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <set>
#include <string>
#include <vector>

#include "src/ReadScanner.h"
#include "utils/Bloom.h"
#include "utils/JChecker.h"
#include "utils/JunctionMap.h"
#include "utils/Kmer.h"

/* Faucet.o is linked too, with its `main` renamed by objcopy (oracle/Makefile), so the sizing KATs go
 * through the reference's OWN my_func (src/Faucet.cpp:197-201) and its globals. */
extern double my_func(double);
extern uint64_t estimated_kmers, singletons;
extern float fpRate;

static kmer_type enc(const std::string& s) {
    kmer_type km = 0;
    getFirstKmerFromRead(&km, const_cast<char*>(&s[0]));
    return km;
}

static void dump_map(const char* name, JunctionMap* jm) {
    printf("{\"kat\":\"scan\",\"name\":\"%s\",\"junctions\":[", name);
    bool first = true;
    for (auto& kv : jm->junctionMap) {
        std::string km(print_kmer(kv.first));
        std::string js = kv.second.toString();
        printf("%s\"%s %s\"", first ? "" : ",", km.c_str(), js.c_str());
        first = false;
    }
    printf("]}\n");
}

static void scan_case(const char* name, int k, int j, int spacer, const std::vector<std::string>& kmers,
                      const std::vector<std::string>& reads) {
    setSizeKmer(k);
    Bloom* b = nullptr;
    b = b->create_bloom_filter_optimal(35, 0.1f);
    b->fakify();
    std::set<bloom_elem> valid;
    for (auto& s : kmers) valid.insert(get_canon(enc(s)));
    b->addFakeKmers(valid);
    JChecker* jc = new JChecker(j, b);
    JunctionMap* jm = new JunctionMap(b, jc, 30);
    Bloom* spf = nullptr; spf = spf->create_bloom_filter_optimal(4, 0.1f);
    Bloom* lpf = nullptr; lpf = lpf->create_bloom_filter_optimal(6, 0.1f);
    ReadScanner sc(jm, "mock_file", b, spf, lpf, jc, spacer);
    for (auto& r : reads) sc.scanInputRead(r, true);
    dump_map(name, jm);
}

/* ref_kat stage3 <bloom file> <bloom size request> <n_hash> <k> <j> <file of hex k-mers>: the reference's own Stage-3 Bloom probes
 * (Bloom::oldContains, JChecker::jcheck(kmer_type), JunctionMap::getValidJExtension, JunctionMap::isBloomJunction) for every k-mer */
static int stage3_kat(int argc, char** argv) {
    if (argc < 8) return 2;
    const int k = atoi(argv[5]), j = atoi(argv[6]);
    setSizeKmer(k);
    Bloom bloom((uint64_t)atoll(argv[3]), k);
    bloom.set_number_of_hash_func(atoi(argv[4]));
    bloom.load(argv[2]);
    JChecker jc(j, &bloom);
    JunctionMap jm(&bloom, &jc, 100);
    FILE* f = fopen(argv[7], "r");
    if (!f) return 2;
    unsigned long long x;
    printf("{\"kat\":\"stage3\",\"k\":%d,\"j\":%d,\"tai\":%llu,\"probes\":[", k, j, (unsigned long long)bloom.tai);
    bool first = true;
    while (fscanf(f, "%llx", &x) == 1) {
        kmer_type km = (kmer_type)x;
        DoubleKmer dk(km);
        printf("%s[\"%llx\",%d,%d,%d,%d]", first ? "" : ",", x, bloom.oldContains(get_canon(km)) ? 1 : 0, jc.jcheck(km) ? 1 : 0,
               jm.getValidJExtension(dk), jm.isBloomJunction(km) ? 1 : 0);
        first = false;
    }
    printf("]}\n");
    fclose(f);
    return 0;
}

/* ref_kat neighbors <bloom file> <bloom size request> <n_hash> <k> <j> <junctions file> <max read length>: the reference's own
 * JunctionMap::findNeighbor (utils/JunctionMap.cpp:231-412) from every junction of a reloaded map along every extension a contig would be
 * built on, on the STATIC map (buildContigGraph removes junctions as it goes; the walk itself is what is pinned here).  One JSON line per
 * call: start k-mer, index, then the result's k-mer, isNode, index, distance, contig length and the contig string itself.  The reference's asserts stay armed: a call
 * that trips one (a filter false positive off the real sequence) is reported as "abort" instead of ending the harness. */
#include <csetjmp>
#include <csignal>
#include <algorithm>
static jmp_buf g_abort_jmp;
static void on_abort(int) { longjmp(g_abort_jmp, 1); }

static int neighbors_kat(int argc, char** argv) {
    if (argc < 9) return 2;
    const int k = atoi(argv[5]), j = atoi(argv[6]);
    setSizeKmer(k);
    Bloom bloom((uint64_t)atoll(argv[3]), k);
    bloom.set_number_of_hash_func(atoi(argv[4]));
    bloom.load(argv[2]);
    JChecker jc(j, &bloom);
    JunctionMap jm(&bloom, &jc, atoi(argv[8]));
    jm.buildFromFile(argv[7]);
    std::vector<kmer_type> keys;
    for (auto& kv : jm.junctionMap) keys.push_back(kv.first);
    std::sort(keys.begin(), keys.end());
    signal(SIGABRT, on_abort);
    FILE* devnull = fopen("/dev/null", "w");
    for (kmer_type key : keys) {
        Junction junc = jm.junctionMap[key];
        for (int i = 0; i < 5; i++) {
            if (junc.dist[i] == 0 || (i < 4 && junc.getCoverage(i) == 0)) continue;
            fflush(stdout);
            if (setjmp(g_abort_jmp)) {
                signal(SIGABRT, on_abort);
                printf("{\"kat\":\"neighbor\",\"start\":\"%llx\",\"index\":%d,\"abort\":1}\n", (unsigned long long)key, i);
                continue;
            }
            BfSearchResult r = jm.findNeighbor(junc, key, i);
            printf("{\"kat\":\"neighbor\",\"start\":\"%llx\",\"index\":%d,\"kmer\":\"%llx\",\"node\":%d,\"rindex\":%d,\"dist\":%d,\"len\":%d,\"contig\":\"%s\"}\n",
                   (unsigned long long)key, i, (unsigned long long)r.kmer, r.isNode ? 1 : 0, r.index, r.distance, (int)r.contig.size(), r.contig.c_str());
        }
    }
    fclose(devnull);
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && std::string(argv[1]) == "stage3") return stage3_kat(argc, argv);
    if (argc > 1 && std::string(argv[1]) == "neighbors") return neighbors_kat(argc, argv);
    /* ---- codec + hash KAT at k=31, tai=2^29 ---- */
    {
        setSizeKmer(31);
        std::string read = "ACGTTGCATGCCGATAGGCTTAACGGATCCAGTCAGGTACCTTGA";
        Bloom b((uint64_t)500000000ULL, 31);
        printf("{\"kat\":\"hash\",\"k\":31,\"tai\":%llu,\"read\":\"%s\",\"pos\":[", (unsigned long long)b.tai, read.c_str());
        for (int p = 0; p + 31 <= (int)read.size(); p++) {
            kmer_type f = getKmerFromRead(read, p);
            kmer_type r = revcomp(f);
            kmer_type c = get_canon(f);
            printf("%s{\"fwd\":\"%016llx\",\"rc\":\"%016llx\",\"hA\":%llu,\"hB\":%llu}", p ? "," : "",
                   (unsigned long long)f, (unsigned long long)r, (unsigned long long)b.oldHash(c, 0),
                   (unsigned long long)b.oldHash(c, 1));
        }
        printf("]}\n");
        /* small k too */
        setSizeKmer(5);
        Bloom b2((uint64_t)1000ULL, 5);
        std::string r2 = "ACGGGCGAACTTTCATAGGA";
        printf("{\"kat\":\"hash\",\"k\":5,\"tai\":%llu,\"read\":\"%s\",\"pos\":[", (unsigned long long)b2.tai, r2.c_str());
        for (int p = 0; p + 5 <= (int)r2.size(); p++) {
            kmer_type f = getKmerFromRead(r2, p);
            printf("%s{\"fwd\":\"%016llx\",\"rc\":\"%016llx\",\"hA\":%llu,\"hB\":%llu}", p ? "," : "",
                   (unsigned long long)f, (unsigned long long)revcomp(f), (unsigned long long)b2.oldHash(get_canon(f), 0),
                   (unsigned long long)b2.oldHash(get_canon(f), 1));
        }
        printf("]}\n");
    }
    /* ---- NT2int ---- */
    {
        const char cs[] = {'A', 'C', 'T', 'G', 'N', 'a'};
        printf("{\"kat\":\"nt2int\",\"vals\":[");
        for (int i = 0; i < 6; i++) printf("%s%d", i ? "," : "", NT2int(cs[i]));
        printf("]}\n");
    }
    /* ---- ctor sizing ---- */
    {
        setSizeKmer(31);
        uint64_t reqs[] = {1000ULL, 1024ULL, 500000ULL, 50000000ULL, 500000000ULL, 5000000000ULL, 1ULL, 7ULL, 8ULL};
        printf("{\"kat\":\"tai\",\"cases\":[");
        for (size_t i = 0; i < sizeof(reqs) / sizeof(reqs[0]); i++) {
            uint64_t req = reqs[i];
            /* the constructor mallocs tai/8 bytes: fine up to 2^33 bits = 1 GiB here */
            Bloom* b = new Bloom(req, 31);
            printf("%s[%llu,%llu]", i ? "," : "", (unsigned long long)req, (unsigned long long)b->tai);
            delete b;
        }
        printf("]}\n");
    }
    /* ---- p1 solver + optimal sizing; silence the reference's own stdout chatter ---- */
    {
        struct C { uint64_t e, s; float fp; };
        C cs[] = {{100000, 20000, .04f}, {10000000, 2000000, .04f}, {100000000, 20000000, .04f}, {1000000000ULL, 200000000ULL, .04f},
                  {100000, 10000, .04f}, {100000, 50000, .04f}, {100000, 80000, .04f}, {100000, 95000, .04f},
                  {2000000000ULL, 1000000000ULL, .04f}, {100000, 20000, .01f}, {100000, 30000, .1f}};
        for (auto& c : cs) {
            uint64_t E = c.e, S = c.s;
            float fp = c.fp;
            estimated_kmers = E;
            singletons = S;
            fpRate = fp;
            std::function<double(double)> f = my_func;
            fflush(stdout);
            FILE* sav = stdout;
            stdout = fopen("/dev/null", "w");
            std::streambuf* cb = std::cout.rdbuf();
            std::cout.rdbuf(nullptr);
            double p1 = brents_fun(f, fp, 0.50, 0.0001, 1000);
            /* the reference's own sizing path (utils/Bloom.cpp:229-247 -> ctor :165-189) */
            Bloom* b = nullptr;
            b = b->create_bloom_filter_optimal(E, p1);
            int nh_eff = b->getNumHash();
            uint64_t tai = b->tai;
            delete b;
            int bits = -1;
            std::cout.rdbuf(cb);
            fclose(stdout);
            stdout = sav;
            (void)bits;
            printf("{\"kat\":\"sizing\",\"E\":%llu,\"S\":%llu,\"fp\":%.9g,\"p1\":%.17g,\"tai\":%llu,\"n_hash\":%d}\n",
                   (unsigned long long)E, (unsigned long long)S, (double)fp, p1, (unsigned long long)tai, nh_eff);
        }
    }
    /* ---- ReadscanTest.cpp cases (k=5, j=0, maxSpacerDist=8, fake Bloom) ---- */
    scan_case("singleReadNoJunctions", 5, 0, 8,
              {"ACGGG", "CGGGC", "GGGCG", "GGCGA", "GCGAA", "CGAAC", "GAACT", "AACTT", "ACTTT", "CTTTC", "TTTCA", "TTCAT", "TCATA",
               "CATAG", "ATAGG", "TAGGA"},
              {"ACGGGCGAACTTTCATAGGA"});
    scan_case("singleReadOneFakeJunction", 5, 0, 8,
              {"ACGGG", "CGGGC", "GGGCG", "GGCGA", "GCGAA", "CGAAC", "GAACT", "AACTT", "AACTC", "ACTCC", "ACTTT", "CTTTC", "TTTCA",
               "TTCAT", "TCATA", "CATAG", "ATAGG", "TAGGA"},
              {"ACGGGCGAACTTTCATAGGA"});
    scan_case("LongReadNoJunctions", 5, 0, 8,
              {"ACGGG", "CGGGC", "GGGCG", "GGCGA", "GCGAA", "CGAAC", "GAACT", "AACTT", "ACTTT", "CTTTC", "TTTCA", "TTCAT", "TCATA",
               "CATAG", "ATAGG", "TAGGA", "AGGAT", "GGATC", "GATCG", "ATCGC", "TCGCA", "CGCAC", "GCACT", "GCACT", "CACTC", "ACTCA",
               "CTCAC"},
              {"ACGGGCGAACTTTCATAGGATCGCACTCAC"});
    scan_case("buildFullMap", 5, 0, 8,
              {"ACGGG", "CGGGC", "GGGCG", "GGCGA", "GCGAA", "CGAAC", "GAACT", "AACTT", "ACTTT", "CTTTC", "TTTCA", "TTCAT", "TCATA",
               "CATAG", "ATAGG", "TAGGA", "GGCGA", "GCGAA", "CGAAC", "GAACT", "AACTA", "ACTAG", "CTAGT", "TAGTC", "AGTCC", "GTCCA",
               "TCCAT", "AACTT", "ACTTT", "CTTTC", "TTTCA", "TTCAT", "TCATA", "CATAC", "ATACG", "TACGA", "ACGAT", "CGATT"},
              {"ACGGGCGAACTTTCATAGGA", "GGCGAACTAGTCCAT", "AACTTTCATACGATT"});
    /* the j=1 variants of the same inputs exercise JChecker::jcheck and the 2j offsets */
    scan_case("buildFullMap_j1", 5, 1, 8,
              {"ACGGG", "CGGGC", "GGGCG", "GGCGA", "GCGAA", "CGAAC", "GAACT", "AACTT", "ACTTT", "CTTTC", "TTTCA", "TTCAT", "TCATA",
               "CATAG", "ATAGG", "TAGGA", "GGCGA", "GCGAA", "CGAAC", "GAACT", "AACTA", "ACTAG", "CTAGT", "TAGTC", "AGTCC", "GTCCA",
               "TCCAT", "AACTT", "ACTTT", "CTTTC", "TTTCA", "TTCAT", "TCATA", "CATAC", "ATACG", "TACGA", "ACGAT", "CGATT"},
              {"ACGGGCGAACTTTCATAGGA", "GGCGAACTAGTCCAT", "AACTTTCATACGATT"});
    scan_case("LongReadNoJunctions_j2_spacer4", 5, 2, 4,
              {"ACGGG", "CGGGC", "GGGCG", "GGCGA", "GCGAA", "CGAAC", "GAACT", "AACTT", "ACTTT", "CTTTC", "TTTCA", "TTCAT", "TCATA",
               "CATAG", "ATAGG", "TAGGA", "AGGAT", "GGATC", "GATCG", "ATCGC", "TCGCA", "CGCAC", "GCACT", "GCACT", "CACTC", "ACTCA",
               "CTCAC"},
              {"ACGGGCGAACTTTCATAGGATCGCACTCAC", "ACGGGCGAACTTTCANAGGATCGCACTCACNNACGGGCGAACT"});
    return 0;
}
