// faucet_binding.cpp — the patch a Faucet maintainer adds to src/Faucet.cpp to run the two hot passes on libfaucet_gpu.so.
//
// Written against the reference's OWN headers (Bloom.h, JunctionMap.h, Junction.h, Kmer.h): tests/test_abi_cpu.py compiles this
// file with `g++ -std=c++11 -fsyntax-only -I/root/reference/src -I/root/reference/utils -Iinclude` wherever the reference tree is
// mounted, so the members it touches (Bloom::tai, ::blooma, ::getNumHash, JunctionMap::createJunction / getJunction,
// Junction::setCoverage / dist / linked) are checked against the real declarations.  It replaces
//     load_two_filters(bloo1, bloo2, read_load_file, fastq, mercy)        src/Faucet.cpp:220   (def utils/Bloom.cpp:267)
//     scanner->scanReads(fastq, paired_ends, no_cleaning)                  src/Faucet.cpp:241-245 (def src/ReadScanner.cpp:284)
// Every fgpu_* status is checked.  No status asks the caller to read its input again: should the scan's preview of the junction walk not
// hold (faucet_gpu.h, fgpu_scan_batch) the library scans its own copy of the batches again -- both inputs may be pipes
// (src/stream_data_from_urls_list.sh feeds Faucet process substitutions).

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "../utils/Bloom.h"
#include "../utils/JunctionMap.h"
#include "../utils/Junction.h"
#include "../utils/Kmer.h"
#include "../utils/JuncPairs.h"
#include "faucet_gpu.h"
#include "pair_loop.h"      // faucet_amd/host: scanReads' paired-end loop over the device's lists, where the device cannot hold the long pair filter's working state
#include "shard_host.h"     // faucet_amd/host: the two passes over several GPUs from this one process (one host thread per device)

extern int j;               // src/Faucet.h:15
extern int maxSpacerDist;   // src/Faucet.h:48

static fgpu_ctx* g_ctx;     // the context Stage 3 asks (with several GPUs: the last shard's, which holds the junction map; every shard holds bloo2)
static faucet_host::ShardedRun* g_run;   // several GPUs: the read shards' contexts and their exchanges
// what the run will ask of the scan, known before pass 1 (the context is made there): set from main() right after handle_arguments,
// e.g. `gpu_configure(!no_cleaning, paired_ends);`.  With cleaning on the scan has to keep scanInputRead's lists (FGPU_FLAG_RECORD_STOPS:
// both pair filters are filled from them on the device); with --paired_ends the lists are kept in any case (the pair counts).
static bool g_want_pairs = false, g_paired_ends = false;
// n_gpus: read shards over that many GPUs (faucet_amd/host/shard_host.h; the reads must then come from regular files).  0 = what the
// environment says (FAUCET_GPUS, FAUCET_TRANSPORT=copy|rccl), else one -- the reference has no such flag; a maintainer adds one to
// handle_arguments (src/Faucet.cpp:57-182) and passes it here.
static int g_gpus = 1, g_transport = FGPU_TRANSPORT_COPY;
void gpu_configure(bool cleaning, bool paired_ends, int n_gpus = 0) {
    g_want_pairs = cleaning;
    g_paired_ends = paired_ends;
    if (n_gpus <= 0) { const char* e = getenv("FAUCET_GPUS"); n_gpus = e ? atoi(e) : 1; }
    g_gpus = n_gpus < 1 ? 1 : n_gpus > 64 ? 64 : n_gpus;
    const char* t = getenv("FAUCET_TRANSPORT");
    g_transport = t && !strcmp(t, "rccl") ? FGPU_TRANSPORT_RCCL : FGPU_TRANSPORT_COPY;
}

static void gpu_die(const char* what, int rc) {
    fprintf(stderr, "%s failed (%d): %s\n", what, rc, g_run && !g_run->error().empty() ? g_run->error().c_str() : fgpu_last_error(g_ctx));
    // fgpu_create leaves a thread of the library setting up the scan's streams; fgpu_destroy joins it, so that exit() does not tear the HIP
    // runtime down under it
    if (g_run) { delete g_run; g_run = NULL; g_ctx = NULL; }
    if (g_ctx) { fgpu_destroy(g_ctx); g_ctx = NULL; }
    exit(2);
}
#define GPU_CHECK(call) do { int rc__ = (call); if (rc__ != FGPU_OK) gpu_die(#call, rc__); } while (0)

// The reference's record loop (utils/Bloom.cpp:280-282,340; src/ReadScanner.cpp:306-308,349), cut into batches of sequence lines.
// `each` returns an fgpu status; the first failure ends the loop and is returned.
template <class F>
static int for_each_batch(const std::string& filename, bool fastq, F each) {
    std::ifstream in(filename.c_str());
    std::string line, bases;
    std::vector<uint64_t> offs(1, 0);
    auto flush = [&]() -> int {
        fgpu_reads r;
        memset(&r, 0, sizeof(r));
        r.bases = bases.data();
        r.offsets = offs.data();
        r.n_reads = offs.size() - 1;
        const int rc = each(&r);
        bases.clear();
        offs.assign(1, 0);
        return rc;
    };
    while (getline(in, line)) {
        line.clear();
        getline(in, line);
        bases += line;
        offs.push_back(bases.size());
        if (fastq) { getline(in, line); getline(in, line); }
        if (offs.size() > 1000000)
            if (int rc = flush()) return rc;
    }
    return offs.size() > 1 ? flush() : FGPU_OK;
}

// replaces the call of load_two_filters in getBloomFilterFromReads() (src/Faucet.cpp:220); bloo1 / bloo2 are the two Bloom objects
// create_bloom_filter_optimal has just made (same tai, same number of hash functions)
void gpu_load_two_filters(Bloom* bloo1, Bloom* bloo2, std::string reads_filename, bool fastq, bool mercy) {
    fgpu_params p;
    memset(&p, 0, sizeof(p));
    p.k = sizeKmer;
    p.j = j;
    p.max_spacer_dist = maxSpacerDist;
    p.n_hash = bloo1->getNumHash();
    p.tai = bloo1->tai;
    p.flags = (mercy ? FGPU_FLAG_MERCY : 0) | ((g_want_pairs || g_paired_ends) ? (FGPU_FLAG_RECORD_STOPS | FGPU_FLAG_KEY_ORDER_FROM_START) : 0);
    if (g_gpus > 1) {
        // read shards: shard r = the r-th file-order share of the records, on device r (fewer devices than shards: they share), one host thread each
        faucet_host::ShardOptions so;
        const int ndev = fgpu_device_count();
        if (ndev < 1) { fprintf(stderr, "fgpu_create failed: no gfx950 device\n"); exit(2); }
        so.n_ranks = g_gpus;
        so.transport = g_transport;
        for (int r = 0; r < g_gpus; r++) so.devices.push_back(r % ndev);
        so.prm = p;
        so.fastq = fastq;
        so.mercy = mercy;
        so.paired_ends = g_paired_ends;
        so.no_cleaning = !g_want_pairs;
        so.chunk_bytes = 32u << 20;
        g_run = new faucet_host::ShardedRun(so);
        int rc = g_run->create();
        if (rc != FGPU_OK) gpu_die("creating the read shards' contexts", rc);
        faucet_host::ShardLoadResult lr;
        if ((rc = g_run->load(reads_filename, &lr)) != FGPU_OK) gpu_die("sharded load pass", rc);
        g_ctx = g_run->last_ctx();     // its bloo1 is the run's (carried-in bits of the lower shards included), as every shard's bloo2 is
        GPU_CHECK(fgpu_bloom_download(g_ctx, FGPU_BLOO1, bloo1->blooma, bloo1->tai / 8));
        GPU_CHECK(fgpu_bloom_download(g_ctx, FGPU_BLOO2, bloo2->blooma, bloo2->tai / 8));
        printf("Reads processed: %llu\nUnambiguous reads: %llu\n", (unsigned long long)lr.stats.reads_processed, (unsigned long long)lr.stats.unambiguous_reads);
        return;
    }
    int rc = fgpu_create(&p, &g_ctx);
    if (rc != FGPU_OK) { fprintf(stderr, "fgpu_create failed (%d): %s\n", rc, fgpu_last_error(NULL)); exit(2); }
    GPU_CHECK(fgpu_load_begin(g_ctx, 0));
    GPU_CHECK(for_each_batch(reads_filename, fastq, [&](const fgpu_reads* r) { return fgpu_load_batch(g_ctx, r); }));
    fgpu_load_stats st;
    GPU_CHECK(fgpu_load_end(g_ctx, &st));
    // the caller deletes bloo1 and keeps bloo2 for the dump and for Stage 3 (src/Faucet.cpp:221-222,262,285-288)
    GPU_CHECK(fgpu_bloom_download(g_ctx, FGPU_BLOO1, bloo1->blooma, bloo1->tai / 8));
    GPU_CHECK(fgpu_bloom_download(g_ctx, FGPU_BLOO2, bloo2->blooma, bloo2->tai / 8));
    printf("Reads processed: %llu\nUnambiguous reads: %llu\n", (unsigned long long)st.reads_processed, (unsigned long long)st.unambiguous_reads);
}

// one pass over the scan file; FGPU_OK, or the status of the first call that failed
static int gpu_scan_pass(const std::string& read_scan_file, bool fastq, fgpu_scan_stats* st) {
    int rc = fgpu_scan_begin(g_ctx);
    if (rc != FGPU_OK) return rc;
    rc = for_each_batch(read_scan_file, fastq, [&](const fgpu_reads* r) { return fgpu_scan_batch(g_ctx, r); });
    const int end_rc = fgpu_scan_end(g_ctx, st);      // always closes the pass, also after a failed batch
    return rc != FGPU_OK ? rc : end_rc;
}

// the device's junction records into the reference's container: creation order -> the reference's unordered_map iteration (= dump) order
static void gpu_fill_junction_map(JunctionMap* junctionMap) {
    uint64_t n = 0;
    GPU_CHECK(fgpu_scan_junction_count(g_ctx, &n));
    std::vector<uint64_t> keys(n ? n : 1);
    std::vector<fgpu_junction> recs(n ? n : 1);
    GPU_CHECK(fgpu_scan_download_junctions(g_ctx, keys.data(), recs.data(), keys.size(), &n));
    for (uint64_t i = 0; i < n; i++) {
        junctionMap->createJunction((kmer_type)keys[i]);
        Junction* jn = junctionMap->getJunction((kmer_type)keys[i]);
        for (int e = 0; e < 4; e++) jn->setCoverage(e, recs[i].cov[e]);
        for (int e = 0; e < 5; e++) { jn->dist[e] = recs[i].dist[e]; jn->linked[e] = recs[i].linked[e] != 0; }
    }
}

// the lines ReadScanner::scanReads and printScanSummary print (src/ReadScanner.cpp:19-27,352-358)
static void gpu_print_scan_summary(const fgpu_scan_stats& st) {
    printf("Reads processed: %llu\nUnambiguous reads: %llu\n", (unsigned long long)st.reads_processed, (unsigned long long)st.unambiguous_reads);
    printf("\nDistinct junctions: %llu \n", (unsigned long long)st.n_junctions);
    printf("Number of kmers that we j-checked: %llu \n", (unsigned long long)st.nb_jcheck_kmer);
    printf("Number of reads with no junctions: %llu \n", (unsigned long long)st.nb_no_juncs);
    printf("Number of processed kmers: %llu \n", (unsigned long long)st.nb_processed);
    printf("Number of skipped kmers: %llu \n", (unsigned long long)st.nb_skipped);
    printf("Reads without errors: %llu\n", (unsigned long long)st.reads_no_errors);
}

// replaces buildJunctionMapFromReads() (src/Faucet.cpp:240-246) for single-end input.  With cleaning on (short_pair_filter != NULL, the
// Bloom made at src/Faucet.cpp:266-283) scan_forward's addPair calls (src/ReadScanner.cpp:208-225) happen on the device and the filter's
// bytes come back at the end; gpu_configure(true, false) must have been called before pass 1 (FGPU_FLAG_RECORD_STOPS).
// several GPUs: the same scan over the read shards; the last shard ends with the run's junction map, pair filters and counters
static void gpu_scan_sharded(JunctionMap* junctionMap, const std::string& read_scan_file, Bloom* short_pair_filter, Bloom* long_pair_filter, bool paired) {
    g_run->set_pair_filters(short_pair_filter ? short_pair_filter->tai : 0, short_pair_filter ? short_pair_filter->getNumHash() : 0,
                            long_pair_filter ? long_pair_filter->tai : 0, long_pair_filter ? long_pair_filter->getNumHash() : 0);
    faucet_host::ShardScanResult sr;
    const int rc = g_run->scan(read_scan_file, &sr);
    if (rc != FGPU_OK) gpu_die("sharded junction scan", rc);
    if (short_pair_filter) GPU_CHECK(fgpu_scan_short_pairs_download(g_ctx, short_pair_filter->blooma, short_pair_filter->tai / 8));
    if (long_pair_filter) {
        uint64_t e = 0, ne = 0;        // (this shard's counts; the run's are the sum over the shards, sr)
        GPU_CHECK(fgpu_scan_long_pairs_download(g_ctx, long_pair_filter->blooma, long_pair_filter->tai / 8, &e, &ne));
    }
    gpu_fill_junction_map(junctionMap);
    if (paired) printf("Empty count: %d, not empty count: %d\n", (int)sr.empty_count, (int)sr.not_empty_count);
    gpu_print_scan_summary(sr.stats);
}

void gpu_scan(JunctionMap* junctionMap, std::string read_scan_file, bool fastq, Bloom* short_pair_filter = NULL) {
    if (g_run) return gpu_scan_sharded(junctionMap, read_scan_file, short_pair_filter, NULL, false);
    fgpu_scan_stats st;
    if (short_pair_filter) GPU_CHECK(fgpu_scan_short_pairs(g_ctx, short_pair_filter->tai, short_pair_filter->getNumHash(), 0));
    const int rc = gpu_scan_pass(read_scan_file, fastq, &st);
    if (rc != FGPU_OK) gpu_die("junction scan", rc);
    if (short_pair_filter) GPU_CHECK(fgpu_scan_short_pairs_download(g_ctx, short_pair_filter->blooma, short_pair_filter->tai / 8));
    gpu_fill_junction_map(junctionMap);
    gpu_print_scan_summary(st);
}

// ---- paired ends (--paired_ends; BASELINE config 3) -------------------------------------------------------------------------------------
// scanReads' paired-end loop (src/ReadScanner.cpp:317-343) keeps the lists of the two ends of a pair and, for every element of the first
// end's list, CHECKS the long pair filter for a partner among the second end's and INSERTS one if there is none -- check-then-insert in file
// order.  Since round 4 that loop runs on the device as well (fgpu_scan_long_pairs: every list element carries its file-order time, the
// decisions are iterated to the fixed point that IS the sequential result), on the lists where they are made: nothing but the finished
// filter's bytes and the two pair counts come back.  The short pair filter is filled on the device as in gpu_scan.
// replaces buildJunctionMapFromReads() for `--paired_ends`: gpu_configure(!no_cleaning, true) before pass 1.  long_pair_filter is the Bloom
// made at src/Faucet.cpp:268-281; short_pair_filter may be NULL (--no_cleaning: the reference then only counts empty / non-empty pairs).
void gpu_scan_paired(JunctionMap* junctionMap, std::string read_scan_file, bool fastq, Bloom* short_pair_filter, Bloom* long_pair_filter,
                     bool no_cleaning) {
    const bool filters = !no_cleaning && long_pair_filter;
    if (g_run) return gpu_scan_sharded(junctionMap, read_scan_file, no_cleaning ? NULL : short_pair_filter, filters ? long_pair_filter : NULL, true);
    if (short_pair_filter && !no_cleaning)
        GPU_CHECK(fgpu_scan_short_pairs(g_ctx, short_pair_filter->tai, short_pair_filter->getNumHash(), 0));
    bool host_loop = false;
    if (filters) {
        const int lrc = fgpu_scan_long_pairs(g_ctx, long_pair_filter->tai, long_pair_filter->getNumHash(), FGPU_LONG_PAIRS_FILTER);
        if (lrc == FGPU_ERR_NOMEM) {
            // The device form of the loop keeps 4 bytes of HBM per filter bit; a filter it cannot hold (--high_cov: E / 2 x 9 bits,
            // src/Faucet.cpp:279-280) is filled HERE, by the reference's own loop (src/ReadScanner.cpp:317-343) over the lists the device hands out
            // (fgpu_scan_take_stops; host/pair_loop.h), straight into long_pair_filter's bit array
            fprintf(stderr, "note: the long pair filter does not fit the device's fixed-point form; the paired-end loop runs on the host\n");
            host_loop = true;
            GPU_CHECK(fgpu_scan_long_pairs(g_ctx, 0, 0, FGPU_LONG_PAIRS_OFF));
            if (short_pair_filter) GPU_CHECK(fgpu_scan_short_pairs(g_ctx, short_pair_filter->tai, short_pair_filter->getNumHash(), 1));   // (lists to the host)
        } else if (lrc != FGPU_OK) {
            gpu_die("fgpu_scan_long_pairs", lrc);
        }
    } else {
        GPU_CHECK(fgpu_scan_long_pairs(g_ctx, 0, 0, FGPU_LONG_PAIRS_COUNT));
    }
    fgpu_scan_stats st;
    uint64_t empty_count = 0, not_empty_count = 0;
    int rc;
    if (host_loop) {
        faucet_host::HostLongPairs hlp((uint8_t*)long_pair_filter->blooma, long_pair_filter->tai, long_pair_filter->getNumHash(), sizeKmer, true);
        std::vector<fgpu_stop> stops;
        std::vector<uint64_t> batch_reads;
        auto take_lists = [&](bool all) -> int {
            for (;;) {
                uint64_t n_stops = 0;
                int64_t seq = -1;
                const int trc = fgpu_scan_take_stops(g_ctx, stops.data(), stops.size(), &n_stops, &seq);
                if (trc == FGPU_ERR_CAPACITY) { stops.resize((size_t)(n_stops + n_stops / 4 + 16)); continue; }
                if (trc != FGPU_OK || seq < 0) return trc;
                hlp.batch(stops.data(), n_stops, batch_reads[(size_t)seq]);
                if (!all) return FGPU_OK;
            }
        };
        rc = fgpu_scan_begin(g_ctx);
        if (rc == FGPU_OK)
            rc = for_each_batch(read_scan_file, fastq, [&](const fgpu_reads* r) -> int {
                const int brc = fgpu_scan_batch(g_ctx, r);
                if (brc != FGPU_OK) return brc;
                batch_reads.push_back(r->n_reads);
                return batch_reads.size() > 1 ? take_lists(false) : FGPU_OK;
            });
        const int end_rc = fgpu_scan_end(g_ctx, &st);
        if (rc == FGPU_OK) rc = end_rc;
        if (rc == FGPU_OK) rc = take_lists(true);
        empty_count = hlp.empty_count;
        not_empty_count = hlp.not_empty_count;
    } else {
        rc = gpu_scan_pass(read_scan_file, fastq, &st);
    }
    if (rc != FGPU_OK) gpu_die("paired-end junction scan", rc);
    if (short_pair_filter && !no_cleaning)
        GPU_CHECK(fgpu_scan_short_pairs_download(g_ctx, short_pair_filter->blooma, short_pair_filter->tai / 8));
    if (!host_loop)
        GPU_CHECK(fgpu_scan_long_pairs_download(g_ctx, filters ? long_pair_filter->blooma : NULL, filters ? long_pair_filter->tai / 8 : 0, &empty_count,
                                                &not_empty_count));
    gpu_fill_junction_map(junctionMap);
    printf("Empty count: %d, not empty count: %d\n", (int)empty_count, (int)not_empty_count);
    gpu_print_scan_summary(st);
}

// ---- Stage 3: JunctionMap::findNeighbor for many (junction, extension) pairs in one device call ------------------------------------------
// getContig (utils/JunctionMap.cpp:133-227) calls findNeighbor(junc, kmer, index) once per contig segment, each call a chain of dependent
// Bloom probes (getValidJExtension, :474-490).  The calls that start from the junctions of the map as it stands are independent of each
// other, so a patched buildBranchingPaths / buildLinearRegions can ask for all of them at once and look the answers up while it builds the
// graph; the results are the BfSearchResult values the reference's function returns (k-mer, isNode, index, distance, contig string).
// `max_read_length` is the JunctionMap constructor's third argument (a private member there).  Calls in which the reference would trip one
// of its asserts come back with kmer == -1, the value of a default-constructed BfSearchResult.
void gpu_set_junction_map(JunctionMap* junctionMap) {
    std::vector<uint64_t> keys;
    std::vector<fgpu_junction> recs;
    for (auto it = junctionMap->junctionMap.begin(); it != junctionMap->junctionMap.end(); ++it) {
        fgpu_junction r;
        memset(&r, 0, sizeof(r));
        for (int e = 0; e < 4; e++) r.cov[e] = it->second.getCoverage(e);
        for (int e = 0; e < 5; e++) { r.dist[e] = it->second.dist[e]; r.linked[e] = it->second.linked[e] ? 1 : 0; }
        keys.push_back((uint64_t)it->first);
        recs.push_back(r);
    }
    GPU_CHECK(fgpu_stage3_set_junctions(g_ctx, keys.data(), recs.data(), keys.size()));
}

std::vector<BfSearchResult> gpu_find_neighbors(const std::vector<kmer_type>& start_kmers, const std::vector<int>& indices, int max_read_length) {
    const size_t n = start_kmers.size();
    std::vector<uint64_t> starts(start_kmers.begin(), start_kmers.end());
    std::vector<int8_t> idx(indices.begin(), indices.end());
    std::vector<fgpu_neighbor> out(n ? n : 1);
    const uint64_t stride = fgpu_stage3_contig_words(sizeKmer, max_read_length);
    std::vector<uint64_t> text((n ? n : 1) * stride);
    GPU_CHECK(fgpu_stage3_find_neighbors(g_ctx, starts.data(), idx.data(), n, max_read_length, out.data(), NULL, text.data(), stride));
    std::vector<BfSearchResult> results(n);
    for (size_t w = 0; w < n; w++) {
        if (out[w].abort) continue;                      // the reference asserts here; the caller sees an unset result
        std::string contig((size_t)out[w].len, 'A');
        for (int b = 0; b < out[w].len; b++) contig[(size_t)b] = "ACTG"[(text[w * stride + (size_t)b / 32] >> (2 * (b % 32))) & 3];
        results[w] = BfSearchResult((kmer_type)out[w].kmer, out[w].node != 0, out[w].rindex, out[w].dist, contig);
    }
    return results;
}
