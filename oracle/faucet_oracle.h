/*
 * faucet_oracle.h — CPU restatement of Faucet's two-pass k-mer pipeline.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle for the MI355X path: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product
 * (libfaucet_gpu.so, the `faucet` CLI) never links or calls anything in oracle/.
 *
 * Parity status: PINNED.  tests/test_oracle_vs_golden.py checks this restatement against
 * fixtures produced by the compiled reference itself (oracle/_ref, built by oracle/Makefile
 * from the sources under /root/reference; generator: tests/golden/make_golden.py) and against
 * the known answers of the reference's own gtest file (src/newTests/ReadscanTest.cpp:102-264).
 *
 * Every function cites the reference file:line whose behaviour it restates.
 */
#ifndef FAUCET_ORACLE_H
#define FAUCET_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- codec (utils/Kmer.cpp) ---------------------------------------------------------- */
int      fo_nt2int(char c);                              /* Kmer.cpp:82-88   */
int      fo_is_valid_nuc(char c);                        /* Kmer.cpp:50-60   */
uint64_t fo_encode(const char* s, int k);                /* Kmer.cpp:429-433 */
uint64_t fo_revcomp(uint64_t x, int k);                  /* Kmer.cpp:238-252 */
uint64_t fo_canon(uint64_t x, int k);                    /* Kmer.cpp:531-533 */
void     fo_decode(uint64_t x, int k, char* out);        /* Kmer.cpp:217 (out gets k chars + NUL) */

/* ---- hashing and sizing (utils/Bloom.h, utils/Bloom.cpp, src/Faucet.cpp) ---------------- */
uint64_t fo_seed(int i);                                 /* Bloom.cpp:500-511, Bloom.h:56-68 */
uint64_t fo_old_hash(uint64_t key, int num, uint64_t tai);       /* Bloom.h:134-145 */
uint64_t fo_bloom_tai(uint64_t requested_bits);          /* Bloom.cpp:165-181 */
/* Brent solver for p1: src/Faucet.cpp:197-209 + utils/Bloom.cpp:33-124.  Returns p1 (or -11). */
double   fo_solve_p1(uint64_t estimated, uint64_t singletons, float fp, int* iterations);
/* create_bloom_filter_optimal: Bloom.cpp:229-247.  n_hash is the value actually in force
 * (4 when the computed value is outside 1..10, Bloom.cpp:491-498). */
void     fo_size_optimal(uint64_t estimated, float fp, int* bits_per_item, uint64_t* tai, int* n_hash);
/* create_bloom_filter_2_hash: Bloom.cpp:206-226 */
void     fo_size_two_hash(uint64_t estimated, float fp, int* bits_per_item, uint64_t* tai, int* n_hash);

/* ---- Bloom object ------------------------------------------------------------------ */
typedef struct fo_bloom fo_bloom;
fo_bloom* fo_bloom_new(uint64_t tai, int n_hash);         /* zeroed bit array of tai/8 bytes */
void      fo_bloom_free(fo_bloom*);
uint8_t*  fo_bloom_bits(fo_bloom*);                       /* blooma, tai/8 bytes */
uint64_t  fo_bloom_nbytes(const fo_bloom*);
float     fo_bloom_weight(const fo_bloom*);               /* Bloom.cpp:191-203 */
void      fo_bloom_fakify(fo_bloom*);                     /* Bloom.cpp:156-158 */
void      fo_bloom_add_fake(fo_bloom*, uint64_t canon);   /* Bloom.cpp:160-162 */
void      fo_bloom_old_add(fo_bloom*, uint64_t canon);    /* Bloom.h:151-159 */
int       fo_bloom_old_contains(fo_bloom*, uint64_t canon);   /* Bloom.h:162-173 */
void      fo_bloom_add_pair(fo_bloom*, uint64_t k1, uint64_t k2, int k);       /* Bloom.cpp:127-139 */
int       fo_bloom_contains_pair(fo_bloom*, uint64_t k1, uint64_t k2, int k);  /* Bloom.cpp:141-154 */
/* counters of single-bit accesses (the T_load / T_scan transaction counts of SURVEY §8d) */
uint64_t  fo_bloom_bit_tests(const fo_bloom*);            /* loads at Bloom.h:252 */
uint64_t  fo_bloom_bit_sets(const fo_bloom*);             /* stores at Bloom.h:224 */
void      fo_bloom_reset_counters(fo_bloom*);

/* ---- read batches ------------------------------------------------------------------ */
/* A batch is `n` sequence lines: line i = bases[offsets[i] .. offsets[i+1]).  Lines are the
 * raw second line of each FASTA/FASTQ record, any byte allowed. */
typedef struct {
    char*     bases;
    uint64_t* offsets;   /* n+1 entries */
    uint64_t  n;
} fo_reads;
/* Record splitting exactly as the reference's getline loops do it
 * (Bloom.cpp:280-282,340; ReadScanner.cpp:306-308,349).  Caller frees with fo_reads_free. */
int  fo_reads_from_file(const char* path, int fastq, fo_reads* out);
void fo_reads_free(fo_reads*);

/* ---- Stage 3's Bloom probes: JChecker::jcheck(kmer_type) (JChecker.cpp:51-80), JunctionMap::getValidJExtension (-1 none, -2 several;
 * JunctionMap.cpp:474-490), JunctionMap::isBloomJunction (:494-504).  kmer = the forward-strand k-mer. */
int fo_stage3_jcheck(fo_bloom* b, uint64_t kmer, int k, int j);
int fo_stage3_valid_extension(fo_bloom* b, uint64_t kmer, int k, int j);
int fo_stage3_bloom_junction(fo_bloom* b, uint64_t kmer, int k, int j);

/* ---- pass 1: load_two_filters (Bloom.cpp:267-299,335-349) ------------------------------ */
typedef struct {
    uint64_t reads_processed;      /* Bloom.cpp:335 */
    uint64_t unambiguous_reads;    /* Bloom.cpp:287 */
    uint64_t kmers;                /* iterations of the loop at Bloom.cpp:289 = the unit N */
    uint64_t to_bloo2;             /* occurrences routed to bloo2 (rho*N) */
} fo_load_stats;
void fo_load_two_filters(fo_bloom* bloo1, fo_bloom* bloo2, const char* bases, const uint64_t* offsets,
                         uint64_t n, int k, fo_load_stats* stats);
/* the same with mercy == true (Bloom.cpp:300-333) */
void fo_load_two_filters_mercy(fo_bloom* bloo1, fo_bloom* bloo2, const char* bases, const uint64_t* offsets,
                         uint64_t n, int k, fo_load_stats* stats);
/* load_single_filter (Bloom.cpp:352-390): unconditional add of every k-mer */
void fo_load_single_filter(fo_bloom* bloo1, const char* bases, const uint64_t* offsets, uint64_t n, int k,
                           fo_load_stats* stats);

/* ---- pass 2: ReadScanner (src/ReadScanner.cpp) ---------------------------------------- */
typedef struct fo_scanner fo_scanner;
typedef struct {
    uint64_t reads_processed;      /* ReadScanner.cpp:348 */
    uint64_t unambiguous_reads;    /* :269 */
    uint64_t reads_no_errors;      /* :276  (= number of valid pieces walked) */
    uint64_t nb_jcheck_kmer;       /* :49  */
    uint64_t nb_no_juncs;          /* :196 */
    uint64_t nb_processed;         /* :83,:192 */
    uint64_t nb_skipped;           /* :192 */
    uint64_t empty_count;          /* :342 */
    uint64_t not_empty_count;      /* :319 */
    uint64_t n_junctions;          /* JunctionMap.cpp:598-600 */
} fo_scan_stats;
/* 14-byte junction record, field order of utils/Junction.h:10-18 */
typedef struct {
    uint8_t cov[4];
    uint8_t dist[5];
    uint8_t linked[5];
} fo_junction;

/* short/long pair filters may be NULL (then the pair bookkeeping of :208-225,:317-343 is skipped
 * exactly when the reference would dereference them; pass no_cleaning=1 in that case). */
fo_scanner* fo_scanner_new(int k, int j, int max_spacer_dist, fo_bloom* bloom,
                           fo_bloom* short_pair_filter, fo_bloom* long_pair_filter);
void        fo_scanner_free(fo_scanner*);
/* scanReads (:284-359) over an in-memory batch; may be called repeatedly (state carries over,
 * including the firstEnd toggle of the paired-end loop). */
void fo_scan_reads(fo_scanner*, const char* bases, const uint64_t* offsets, uint64_t n,
                   int paired_ends, int no_cleaning);
/* scanInputRead (:260-282) on one line; returns the number of junction extensions it yields and
 * copies up to cap of them into ext_out. */
uint64_t fo_scan_input_read(fo_scanner*, const char* line, uint64_t len, int no_cleaning,
                            uint64_t* ext_out, uint64_t cap);
/* the same, with what the callers of that list need to know about every visit (the layout of fgpu_stop::info in include/faucet_gpu.h:
 * ReadKmer::pos inside the valid piece, bit 28 facing FORWARD, bit 29 first element of its scan_forward call, bit 30 the fake junction's) */
uint64_t fo_scan_input_read_ex(fo_scanner*, const char* line, uint64_t len, int no_cleaning,
                               uint64_t* ext_out, uint32_t* info_out, uint64_t cap);
void     fo_scan_get_stats(const fo_scanner*, fo_scan_stats*);
/* Junction map contents.  order = 0: iteration order of the std::unordered_map (= dump order of
 * JunctionMap::writeToFile, JunctionMap.cpp:588-593); order = 1: creation order. */
uint64_t fo_scan_get_junctions(const fo_scanner*, int order, uint64_t* keys, fo_junction* recs, uint64_t cap);
/* test-only (no reference counterpart): continue from a junction map and counters produced elsewhere */
void fo_scan_import(fo_scanner*, const uint64_t* keys, const fo_junction* recs, uint64_t n, const fo_scan_stats* carried);
/* of the filter's bit tests (fo_bloom_bit_tests), how many were made by getValidReads (:233-257) */
uint64_t fo_scan_bit_tests_valid(const fo_scanner*);
/* writeToFile (JunctionMap.cpp:579-596) — the .junctions text format of Junction.cpp:74-89 */
int      fo_scan_write_junctions(const fo_scanner*, const char* path);

/* valid pieces of one unambiguous segment, getValidReads (:233-257): writes (start,len) pairs */
uint64_t fo_get_valid_reads(fo_scanner*, const char* seg, uint64_t len, uint64_t* start_len_out, uint64_t cap);
/* testForJunction (:36-56) at half-step t of piece; returns flag, *njcheck gets the NbJCheckKmer increment */
int      fo_test_for_junction(fo_scanner*, const char* piece, uint64_t len, int t, int* njcheck);

#ifdef __cplusplus
}
#endif
#endif
