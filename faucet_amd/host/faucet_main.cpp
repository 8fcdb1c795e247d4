// faucet_main.cpp — the `faucet` command line on top of libfaucet_gpu.so (C ABI, include/faucet_gpu.h).
//
// Host side of the hot path in the reference's own language (C++11).  Same flags, same required arguments, same
// exit codes and the same output files as src/Faucet.cpp:57-182,248-300 for the part of the run this repository
// covers: the Bloom load pass (-> <prefix>.bloom) and the junction scan (-> <prefix>.junctions, and without
// --no_cleaning <prefix>.short_pair_filter / .long_pair_filter).  The contig-graph stage is not part of this build:
// the program stops after the last file the scan produces (exit code 3 when the reference would have gone on to the
// contig graph; the reference itself can be restarted from these files with -bloom_file / -junctions_file).
// No k-mer window, filter probe or junction is computed on the host: everything goes through fgpu_*; if the library
// finds no gfx950 device the program fails (there is no CPU path).  What stays on the host is what SURVEY.md 8(b)
// leaves there: the JunctionMap container that fixes the dump order, and the files.  Both pair filters are filled on the
// device from scanInputRead's per-read lists (fgpu_scan_short_pairs, fgpu_scan_long_pairs) and come back as bytes.
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <fstream>
#include <mutex>
#include <thread>
#include <string>
#include <unordered_map>
#include <vector>

#include "faucet_gpu.h"
#include "junction_order.h"
#include "pair_loop.h"
#include "shard_host.h"
#include "text_source.h"

using faucet_host::TextSource;
using faucet_host::pinned_slot;
using faucet_host::kTextPad;

namespace {

struct Options {   // globals of src/Faucet.h:14-53
    float fp_rate = .04f;
    int j = 1;
    std::string read_load_file, read_scan_file, bloom_input_file, junctions_prefix, file_prefix;
    int read_length = 0, k = 0;
    uint64_t estimated_kmers = 0, singletons = 0;
    bool load_file_flag = false, scan_file_flag = false, k_val_flag = false, max_len_flag = false, est_kmers_flag = false,
         est_sing_flag = false, pref_flag = false;
    bool two_hash = false, from_bloom = false, from_junctions = false, just_load = false, fastq = false, mercy = false,
         node_graph = false, paired_ends = false, no_cleaning = false, high_cov = false;
    int max_spacer_dist = 100;
    uint64_t batch_reads = 0;         // not a reference flag: > 0 = split records on the host, this many reads per device call
    uint64_t chunk_mb = 64;           // not a reference flag: file text handed to the device per call, records split there
    uint64_t chunk_bytes = 0;         // = chunk_mb << 20, less for regular files that are smaller (main)
    int gpus = 1;                     // not a reference flag: read shards over this many GPUs, one host thread each (shard_host.h)
    std::string transport = "copy";   // not a reference flag: how the shards' bitmaps and tables travel: copy (device-to-device copies) | rccl
};

void argument_error() {   // src/Faucet.cpp:50-54
    fprintf(stderr, "Usage:\n");
    fprintf(stderr, "./faucet -read_load_file <filename> -read_scan_file <filename> -size_kmer <k> -max_read_length <length> "
                    "-estimated_kmers <num_kmers> -singletons <num_kmers> -file_prefix <prefix>");
    fprintf(stderr, "\nOptional arguments: --fastq --mercy --high_cov -max_spacer_dist <dist> -fp rate <rate> -j <int> --two_hash "
                    "-bloom_file <filename> -junctions_file <filename> --paired_ends --no_cleaning\n");
}

// src/Faucet.cpp:57-182 (with the missing `return 0` supplied and a bounds check on flag values)
int handle_arguments(int argc, char** argv, Options& o) {
    if (argc == 1) { argument_error(); return 1; }
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto val = [&](const char*& out) { if (i + 1 >= argc) return false; out = argv[++i]; return true; };
        const char* v = nullptr;
        if (a == "-read_load_file") { if (!val(v)) goto bad; o.read_load_file = v; o.load_file_flag = true; }
        else if (a == "-read_scan_file") { if (!val(v)) goto bad; o.read_scan_file = v; o.scan_file_flag = true; }
        else if (a == "-size_kmer") { if (!val(v)) goto bad; o.k = atoi(v); o.k_val_flag = true; }
        else if (a == "-max_read_length") { if (!val(v)) goto bad; o.read_length = atoi(v); o.max_len_flag = true; }
        else if (a == "-estimated_kmers") { if (!val(v)) goto bad; o.estimated_kmers = (uint64_t)atoll(v); o.est_kmers_flag = true; }
        else if (a == "-singletons") { if (!val(v)) goto bad; o.singletons = (uint64_t)atoll(v); o.est_sing_flag = true; }
        else if (a == "-fp") { if (!val(v)) goto bad; o.fp_rate = (float)atof(v); }
        else if (a == "-j") { if (!val(v)) goto bad; o.j = atoi(v); }
        else if (a == "-file_prefix") { if (!val(v)) goto bad; o.file_prefix = v; o.pref_flag = true; }
        else if (a == "--two_hash") o.two_hash = true;
        else if (a == "--just_load_bloom") o.just_load = true;
        else if (a == "--no_cleaning") o.no_cleaning = true;
        else if (a == "--fastq") o.fastq = true;
        else if (a == "--mercy") o.mercy = true;
        else if (a == "--high_cov") o.high_cov = true;
        else if (a == "--node_graph") o.node_graph = true;
        else if (a == "--paired_ends") o.paired_ends = true;
        else if (a == "-bloom_file") { if (!val(v)) goto bad; o.bloom_input_file = v; o.from_bloom = true; }
        else if (a == "-max_spacer_dist") { if (!val(v)) goto bad; o.max_spacer_dist = atoi(v); }
        else if (a == "-junctions_file") { if (!val(v)) goto bad; o.junctions_prefix = v; o.from_junctions = true; }
        else if (a == "-batch_reads") { if (!val(v)) goto bad; o.batch_reads = (uint64_t)atoll(v); }
        else if (a == "-chunk_mb") { if (!val(v)) goto bad; o.chunk_mb = (uint64_t)atoll(v); if (!o.chunk_mb) o.chunk_mb = 1; }
        else if (a == "-gpus") { if (!val(v)) goto bad; o.gpus = atoi(v); if (o.gpus < 1 || o.gpus > 64) { fprintf(stderr, "-gpus must be in 1..64\n"); return 1; } }
        else if (a == "-transport") { if (!val(v)) goto bad; o.transport = v; if (o.transport != "copy" && o.transport != "rccl") { fprintf(stderr, "-transport must be copy or rccl\n"); return 1; } }
        else if (a == "--help" || a == "-h") { argument_error(); return 1; }
        else { fprintf(stderr, "Cannot parse tag %s\n", argv[i]); argument_error(); return 1; }
        continue;
    bad:
        fprintf(stderr, "Missing value after %s\n", a.c_str());
        argument_error();
        return 1;
    }
    if (!(o.load_file_flag && o.scan_file_flag && o.k_val_flag && o.max_len_flag && o.est_kmers_flag && o.est_sing_flag && o.pref_flag)) {
        fprintf(stderr, "Some required argument is missing.\n");
        argument_error();
        return 1;
    }
    if (o.from_junctions && !o.from_bloom) {
        fprintf(stderr, "Cannot start from junctions without a bloom file.\n");
        argument_error();
        return 1;
    }
    if (o.from_junctions) printf("Starting from after read scan based on bloom and junction files.\n");
    else if (o.from_bloom) printf("Starting from after bloom load based on bloom file.\n");
    else printf("Starting at the beginning: will load bloom and find junctions from the read set.\n");
    if (o.just_load) printf("Only loading bloom, dumping and termination.\n");
    printf(o.node_graph ? "Using node graph.\n" : "Using contig graph.\n");          // src/Faucet.cpp:146-149 (a switch of the stage this build stops before)
    printf("Read load file name: %s\n", o.read_load_file.c_str());
    printf("Read scan file name: %s\n", o.read_scan_file.c_str());
    printf("k: %d \n", o.k);
    printf("Maximal read length: %d\n", o.read_length);
    printf("Estimated number of distinct kmers, for sizing bloom filter: %llu.\n", (unsigned long long)o.estimated_kmers);
    printf("False positive rate: %f\n", o.fp_rate);
    printf("File prefix: %s\n", o.file_prefix.c_str());
    printf("Max spacer dist: %d\n", o.max_spacer_dist);
    printf(o.two_hash ? "Using 2 hash functions.\n" : "Using space-optimal hash settings.\n");
    printf("Paired ends: %d\n", (int)o.paired_ends);
    // src/Faucet.cpp:177-181 prints sizeof(Junction / ContigNode / Contig / int / long) of ITS build: the reference's values on x86-64 Linux,
    // so that what a caller greps or diffs in the log stays where it is
    printf("Size of junction: 14\nSize of contigNode: 48\nSize of contig: 104\nSize of int: 4\nSize of long: 8\n");
    return 0;
}

// Record splitting exactly as the reference's loops do it (utils/Bloom.cpp:280-282,340; src/ReadScanner.cpp:306-308,349):
//   while (getline(header)) { getline(sequence); ...; if (fastq) getline, getline; }
// Works on non-seekable input (the reference is fed process substitutions, src/stream_data_from_urls_list.sh:12-15).
class ReadSource {
public:
    ReadSource(const std::string& path, bool fastq) : in_(path.c_str()), fastq_(fastq) {}
    bool is_open() const { return in_.is_open(); }
    // next batch of at most max_reads sequence lines; false when the input is exhausted
    bool next(uint64_t max_reads, std::vector<char>& bases, std::vector<uint64_t>& offsets) {
        bases.clear();
        offsets.assign(1, 0);
        std::string line;
        while (offsets.size() - 1 < max_reads && std::getline(in_, line)) {
            line.clear();
            std::getline(in_, line);
            bases.insert(bases.end(), line.begin(), line.end());
            offsets.push_back(bases.size());
            if (fastq_) { std::getline(in_, line); std::getline(in_, line); }
        }
        return offsets.size() > 1;
    }
private:
    std::ifstream in_;
    bool fastq_;
};

// one interface over both ways of cutting the input into batches
class BatchSource {
public:
    BatchSource(const Options& o, const std::string& path)
        : host_(o.batch_reads ? new ReadSource(path, o.fastq) : nullptr),
          text_(o.batch_reads ? nullptr : new TextSource(path, o.fastq, o.chunk_bytes)), batch_reads_(o.batch_reads) {}
    ~BatchSource() { delete host_; delete text_; }
    bool is_open() const { return host_ ? host_->is_open() : text_->is_open(); }
    int next(fgpu_ctx* ctx, fgpu_reads* out) {
        if (text_) return text_->next(ctx, out);
        if (!host_->next(batch_reads_, bases_, offsets_)) return 0;
        fgpu_reads r = {bases_.data(), offsets_.data(), offsets_.size() - 1, 0, 0, nullptr};
        *out = r;
        return 1;
    }
private:
    ReadSource* host_;
    TextSource* text_;
    uint64_t batch_reads_;
    std::vector<char> bases_;
    std::vector<uint64_t> offsets_;
};

#define CHECK(call)                                                                   \
    do {                                                                              \
        int rc__ = (call);                                                            \
        if (rc__ != FGPU_OK) {                                                        \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc__, fgpu_last_error(ctx)); \
            return 2;                                                                 \
        }                                                                             \
    } while (0)

struct Junction {   // utils/Junction.h:10-18
    uint8_t cov[4], dist[5], linked[5];
};

const char kDecode[4] = {'A', 'C', 'T', 'G'};   // utils/Kmer.cpp:21

// ---- the pair filters: sized and written by the host, filled on the device ---------------------------------------
struct PairFilter {   // a Bloom used through addPair / containsPair only (utils/Bloom.cpp:127-154): here just its bytes, which the device fills
    uint64_t tai = 0;
    int n_hash = 0;
    std::vector<uint8_t> bits;
    void create(uint64_t elements, float fp) {   // create_bloom_filter_optimal, utils/Bloom.cpp:229-247
        int32_t bpk = 0, nh = 0;
        fgpu_size_optimal(elements, fp, &bpk, &tai, &nh);
        n_hash = nh;
        printf("Bits per kmer: %d \n", bpk);
        printf("BF memory: %f MB\n", (float)((elements * (uint64_t)bpk) / 8ULL / 1024ULL) / 1024);
        printf("Number of hash functions: %d \n", n_hash);
        bits.assign(tai / 8, 0);
    }
    float weight() const {   // Bloom::weight, utils/Bloom.cpp:191-203
        long w = 0;
        size_t i = 0;
        for (; i + 8 <= bits.size(); i += 8) {       // a word at a time (the filter of a --no_cleaning run is 8 MiB of zeros: 10 ms byte by byte)
            uint64_t x;
            memcpy(&x, &bits[i], 8);
            w += __builtin_popcountll(x);
        }
        for (; i < bits.size(); i++) w += __builtin_popcount(bits[i]);
        return (float)w / (float)tai;
    }
    int dump(const std::string& path) const {   // Bloom::dump, utils/Bloom.cpp:571-578
        FILE* f = fopen(path.c_str(), "wb");
        if (!f) { fprintf(stderr, "cannot write %s\n", path.c_str()); return 2; }
        const bool ok = fwrite(bits.data(), 1, bits.size(), f) == bits.size();
        if (fclose(f) != 0 || !ok) { fprintf(stderr, "cannot write %s\n", path.c_str()); return 2; }
        return 0;
    }
};

// The junction container of the reference, for the paths that need a real one: -junctions_file (a later line with the same k-mer replaces the
// earlier one) and the dump order on a standard library whose container DumpOrder does not replay (junction_order.h).
typedef std::unordered_map<uint64_t, Junction> JunctionTable;

// JunctionMap::writeToFile (utils/JunctionMap.cpp:579-596), Junction::toString (utils/Junction.cpp:74-89):
//   "<kmer> d0 d1 d2 d3 d4  c0 c1 c2 c3 <sum>  l0 l1 l2 l3 l4 \n"
// in the order given (indices into keys / recs), formatted by hand: a million lines, a few threads, each its own stretch of the order.
int write_junctions(const std::string& path, const uint64_t* keys, const fgpu_junction* recs, const std::vector<uint32_t>& order, int k) {
    const int fd = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (fd < 0) { fprintf(stderr, "cannot write %s\n", path.c_str()); return 2; }
    const size_t kLine = (size_t)k + 1 + 5 * 4 + 1 + 4 * 4 + 5 + 1 + 5 * 2 + 1;   // longest line: three digits and a blank per byte field
    auto format = [&](size_t from, size_t to, std::vector<char>& buf) -> size_t {      // returns the bytes written into buf (sized once per thread, never re-zeroed)
        if (buf.size() < (to - from) * kLine) buf.resize((to - from) * kLine);
        char* w = buf.data();
        auto put_uint = [&](unsigned v) {
            if (v >= 1000) *w++ = (char)('0' + v / 1000);
            if (v >= 100) *w++ = (char)('0' + v / 100 % 10);
            if (v >= 10) *w++ = (char)('0' + v / 10 % 10);
            *w++ = (char)('0' + v % 10);
            *w++ = ' ';
        };
        for (size_t at = from; at < to; at++) {
            if (at + 8 < to) { __builtin_prefetch(&keys[order[at + 8]]); __builtin_prefetch(&recs[order[at + 8]]); }
            uint64_t x = keys[order[at]];
            for (int i = k - 1; i >= 0; i--) { w[i] = kDecode[x & 3]; x >>= 2; }
            w += k;
            *w++ = ' ';
            const fgpu_junction& j = recs[order[at]];
            for (int i = 0; i < 5; i++) put_uint(j.dist[i]);
            *w++ = ' ';
            for (int i = 0; i < 4; i++) put_uint(j.cov[i]);
            const unsigned sum = (unsigned)j.cov[0] + j.cov[1] + j.cov[2] + j.cov[3];
            put_uint(sum);   // at most 4 * 255
            *w++ = ' ';
            for (int i = 0; i < 5; i++) { *w++ = j.linked[i] ? '1' : '0'; *w++ = ' '; }
            *w++ = '\n';
        }
        return (size_t)(w - buf.data());
    };
    // Round 6 (profiles/r06_cli_large.txt: 1.25 s for config 5's 2.95e7 lines, formatting and writing in turn): the order is cut into chunks that
    // the threads take one after the other; a chunk's place in the file is known as soon as every chunk before it has been FORMATTED (its size), and
    // it is then written there (pwrite) while other chunks are still being formatted or written -- formatting and writing overlap, both in parallel.
    const size_t kChunk = getenv("FAUCET_DEBUG_DUMP_CHUNK") ? (size_t)std::max(1, atoi(getenv("FAUCET_DEBUG_DUMP_CHUNK"))) : (size_t)1 << 18;   // (tests: many small chunks)
    const size_t n_chunks = (order.size() + kChunk - 1) / kChunk;
    const size_t n_threads = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(8, std::thread::hardware_concurrency() ? std::thread::hardware_concurrency() : 4), n_chunks));
    std::vector<uint64_t> offset(n_chunks + 1, 0);
    std::mutex m;
    std::condition_variable cv;
    size_t known = 0, next = 0;          // offset[0 .. known] are final; next chunk to hand out
    bool failed = false;
    auto worker = [&]() {
        std::vector<char> buf;
        for (;;) {
            size_t i;
            { std::lock_guard<std::mutex> g(m); if (next >= n_chunks || failed) return; i = next++; }
            const size_t bytes = format(i * kChunk, std::min(order.size(), (i + 1) * kChunk), buf);
            uint64_t at;
            {
                std::unique_lock<std::mutex> g(m);
                cv.wait(g, [&] { return known == i || failed; });      // (chunks are handed out in order: chunk i - 1 is being formatted or done)
                if (failed) return;
                at = offset[i];
                offset[i + 1] = at + bytes;
                known = i + 1;
            }
            cv.notify_all();
            size_t done = 0;
            while (done < bytes) {
                const ssize_t w = pwrite(fd, buf.data() + done, bytes - done, (off_t)(at + done));
                if (w <= 0) { std::lock_guard<std::mutex> g(m); failed = true; break; }
                done += (size_t)w;
            }
            if (failed) { cv.notify_all(); return; }
        }
    };
    std::vector<std::thread> th;
    for (size_t t = 1; t < n_threads; t++) th.emplace_back(worker);
    worker();
    for (std::thread& t : th) t.join();
    const bool bad = failed || close(fd) != 0;
    if (bad) { fprintf(stderr, "cannot write %s\n", path.c_str()); return 2; }
    return 0;
}

// JunctionMap::buildFromFile (utils/JunctionMap.cpp:619-639) with Junction's parsing constructor (utils/Junction.cpp:102-118): the k-mer as
// text (getFirstKmerFromRead: A0 C1 T2 G3), five distances, four coverages, their sum (skipped), five link flags; a later line with the same
// k-mer replaces the earlier one (junctionMap[kmer] = junc).  Returns -1 when the file cannot be opened or a line does not parse.
long read_junctions(const std::string& path, int k, JunctionTable& map) {
    std::ifstream in(path.c_str());
    if (!in.is_open()) return -1;
    std::string line;
    long n_lines = 0;
    while (std::getline(in, line)) {
        if ((int)line.size() < k + 1) return -1;
        uint64_t kmer = 0;
        for (int i = 0; i < k; i++) {
            const char c = line[(size_t)i];
            if (c != 'A' && c != 'C' && c != 'G' && c != 'T') return -1;
            kmer = (kmer << 2) | (uint64_t)((c >> 1) & 3);
        }
        int v[15];
        const char* p = line.c_str() + k;
        for (int i = 0; i < 15; i++) {
            char* end = nullptr;
            const long x = strtol(p, &end, 10);
            if (end == p) return -1;
            v[i] = (int)x;
            p = end;
        }
        Junction j;
        for (int i = 0; i < 5; i++) j.dist[i] = (uint8_t)v[i];
        for (int i = 0; i < 4; i++) j.cov[i] = (uint8_t)v[5 + i];
        for (int i = 0; i < 5; i++) j.linked[i] = (uint8_t)(v[10 + i] != 0);
        map[kmer] = j;
        n_lines++;
    }
    return n_lines;
}

int load_pair_filter(PairFilter& pf, const std::string& path) {   // Bloom::load (utils/Bloom.cpp:580-587): what fits is taken, like -bloom_file
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path.c_str()); return 2; }
    printf("loading bloom filter from file, nelem %llu \n", (unsigned long long)pf.bits.size());
    const size_t got = fread(pf.bits.data(), 1, pf.bits.size(), f);
    const bool more = fgetc(f) != EOF;
    fclose(f);
    if (got != pf.bits.size() || more)
        fprintf(stderr, "note: %s is not of the %llu bytes this run's filter has: its head is taken, what it does not cover stays empty (as in the reference)\n",
                path.c_str(), (unsigned long long)pf.bits.size());
    printf("bloom loaded\n");
    return 0;
}

}  // namespace

// FGPU_CLI_TIMES=1: phase clock on stderr (measurement aid; stdout stays what the reference prints)
struct PhaseClock {
    bool on = getenv("FGPU_CLI_TIMES") != nullptr;
    double batch_ms = 0, take_ms = 0, pairs_ms = 0, scan_ms = 0;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(), last = t0;
    void mark(const char* what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[cli] %-28s %9.2f ms  (at %9.2f ms)\n", what, std::chrono::duration<double, std::milli>(now - last).count(),
                std::chrono::duration<double, std::milli>(now - t0).count());
        last = now;
    }
};

// What both kinds of run -- one device, or read shards over several -- do once the scan has ended: the lines ReadScanner::scanReads and
// printScanSummary print (src/ReadScanner.cpp:19-27,352-358), the junction map (held by `ctx`: the only context, or the last shard's) into
// <prefix>.junctions in the reference's dump order, and the pair filter files (src/Faucet.cpp:296-300).  0, or the exit code of a failure.
int write_scan_outputs(const Options& o, fgpu_ctx* ctx, const fgpu_scan_stats& ss, uint64_t empty_count, uint64_t not_empty_count, double seconds,
                       const PairFilter& short_pf, const PairFilter& long_pf, PhaseClock& clk) {
    printf("Empty count: %d, not empty count: %d\n", (int)empty_count, (int)not_empty_count);
    printf("Reads processed: %llu\n", (unsigned long long)ss.reads_processed);
    printf("Unambiguous reads: %llu\n", (unsigned long long)ss.unambiguous_reads);
    printf("Time in seconds for read scan: %f \n", seconds);
    printf("\nDistinct junctions: %llu \n", (unsigned long long)ss.n_junctions);
    printf("Number of kmers that we j-checked: %llu \n", (unsigned long long)ss.nb_jcheck_kmer);
    printf("Number of reads with no junctions: %llu \n", (unsigned long long)ss.nb_no_juncs);
    printf("Number of processed kmers: %llu \n", (unsigned long long)ss.nb_processed);
    printf("Number of skipped kmers: %llu \n", (unsigned long long)ss.nb_skipped);
    printf("Reads without errors: %llu\n", (unsigned long long)ss.reads_no_errors);

    // junction records come back in creation order; the reference's container gives the reference's dump order
    uint64_t n = 0;
    CHECK(fgpu_scan_junction_count(ctx, &n));
    std::vector<uint64_t> keys(n ? n : 1);
    std::vector<fgpu_junction> recs(n ? n : 1);
    CHECK(fgpu_scan_download_junctions(ctx, keys.data(), recs.data(), keys.size(), &n));
    clk.mark("  junction download");
    // the reference's dump order = the iteration order of its container after these insertions (junction_order.h).  Round 6: computed on the DEVICE
    // from the closed form of what the container does to its node list (fgpu_scan_dump_order: one radix sort per stretch between rehashes; the host
    // replay was 5.7 s of a 9.5 s run at config 5's 2.95e7 junctions) -- after the device has shown, on a prefix of these very keys, that it agrees
    // with the host's replay, which has shown that it agrees with a real container.  Anything else falls back to the replay, or to the container.
    std::vector<uint32_t> order;
    const size_t prefix = (size_t)std::min<uint64_t>(n, 50000);
    const bool replay_ok = DumpOrder::agrees_with_the_container(keys.data(), prefix);
    bool on_device = false;
    if (replay_ok && n) {
        auto device_order = [&](size_t m, std::vector<uint32_t>* out) -> bool {
            const std::vector<DumpOrder::Rehash> sch = DumpOrder::schedule(m);
            std::vector<uint64_t> counts, buckets;
            for (const DumpOrder::Rehash& r : sch) { counts.push_back(r.count); buckets.push_back(r.buckets); }
            out->assign(m, 0);
            return fgpu_scan_dump_order(ctx, counts.data(), buckets.data(), counts.size(), m, out->data()) == FGPU_OK;
        };
        std::vector<uint32_t> head;
        if (device_order(prefix, &head) && head == DumpOrder::of(keys.data(), prefix) && device_order((size_t)n, &order)) on_device = true;
        if (on_device && getenv("FAUCET_DEBUG_DUMP_ORDER_CHECK")) {     // (tests, measurements: the whole order against the host replay)
            const bool same = order == DumpOrder::of(keys.data(), (size_t)n);
            fprintf(stderr, "[cli] dump order of %llu junctions on the device %s the host replay's\n", (unsigned long long)n, same ? "equals" : "DIFFERS FROM");
            if (!same) return 2;
        }
    }
    if (!on_device) {
        if (replay_ok) {
            order = DumpOrder::of(keys.data(), (size_t)n);
        } else {   // a standard library that links its nodes another way: ask the container itself
            order.clear();
            std::unordered_map<uint64_t, uint32_t> container;
            for (uint64_t i = 0; i < n; i++) container.insert(std::pair<uint64_t, uint32_t>(keys[i], (uint32_t)i));
            for (const auto& kv : container) order.push_back(kv.second);
        }
    }
    if (clk.on) fprintf(stderr, "[cli]   dump order %s\n", on_device ? "on the device (fgpu_scan_dump_order)" : replay_ok ? "by the host replay" : "by the container");
    clk.mark("  dump order");
    // JunctionMap::writeToFile (utils/JunctionMap.cpp:579-596) first counts the junctions that are solid at thresholds 0..4
    // (Junction::isSolid, utils/Junction.cpp:38-46: more than one of the four extensions with that much coverage)
    for (int thr = 0; thr < 5; thr++) {
        long solid = 0;
        for (uint64_t i = 0; i < n; i++) {
            int paths = 0;
            for (int e = 0; e < 4; e++) paths += recs[i].cov[e] >= thr ? 1 : 0;
            solid += paths > 1 ? 1 : 0;
        }
        printf("There are %ld junctions with solidity at least %d.\n", solid, thr);
    }
    printf("Writing to junction file\n");
    if (int rc = write_junctions(o.file_prefix + ".junctions", keys.data(), recs.data(), order, o.k)) return rc;
    printf("Done writing to junction file\n");
    clk.mark("junction download + dump");
    if (!o.no_cleaning) {   // src/Faucet.cpp:297-300
        if (int rc = short_pf.dump(o.file_prefix + ".short_pair_filter")) return rc;
        printf("bloom dumped \n");                                   // (Bloom::dump's line, utils/Bloom.cpp:571-578: once per filter written)
        if (o.paired_ends) {
            if (int rc = long_pf.dump(o.file_prefix + ".long_pair_filter")) return rc;
            printf("bloom dumped \n");
        }
    }
    printf("Weight of short pair filter: %f\n", short_pf.weight());
    if (o.paired_ends) printf("Weight of long pair filter: %f\n", long_pf.weight());
    printf("Number of junctions: %llu\n", (unsigned long long)order.size());
    return 0;
}

// ---- `-gpus N`: the same run with the reads sharded over N GPUs, one host thread per device (shard_host.h) -------------------------------
// Same files, same log, same exit codes as the single-device run below; what differs is who does the passes: shard r = the r-th file-order
// share of the records, on device r (with fewer devices than shards several shards share a device: what a one-GPU box can test).  Inputs
// must be regular files (a shard is a byte range); -bloom_file restarts load the filter on every device.
int main_sharded(const Options& o, const fgpu_params& prm, uint64_t tai, PhaseClock& clk) {
    using namespace faucet_host;
    if (o.batch_reads) { fprintf(stderr, "-batch_reads (records split on the host) cannot be combined with -gpus: the shards split their records on their devices\n"); return 1; }
    for (const std::string* f : {&o.read_load_file, &o.read_scan_file}) {      // (before any device is touched)
        struct stat st;
        if (o.from_bloom && f == &o.read_load_file) continue;
        if (stat(f->c_str(), &st) != 0) { fprintf(stderr, "cannot open %s\n", f->c_str()); return 2; }
        if (!S_ISREG(st.st_mode)) { fprintf(stderr, "%s is not a regular file: with -gpus the read shards are byte ranges of their input (pipes need -gpus 1)\n", f->c_str()); return 2; }
    }
    const int ndev = fgpu_device_count();
    if (ndev < 1) { fprintf(stderr, "fgpu_create failed (%d): no gfx950 device\n", FGPU_ERR_HIP); return 2; }
    ShardOptions so;
    so.n_ranks = o.gpus;
    so.transport = o.transport == "rccl" ? FGPU_TRANSPORT_RCCL : FGPU_TRANSPORT_COPY;
    for (int r = 0; r < o.gpus; r++) so.devices.push_back(r % ndev);
    if (ndev < o.gpus) fprintf(stderr, "note: %d read shards on %d device%s: shards share devices\n", o.gpus, ndev, ndev == 1 ? "" : "s");
    so.prm = prm;
    so.fastq = o.fastq;
    so.mercy = o.mercy;
    so.paired_ends = o.paired_ends;
    so.no_cleaning = o.no_cleaning;
    so.verbose = clk.on;
    // every shard reads its share through buffers of its own: chunks no larger than a share needs
    so.chunk_bytes = o.chunk_bytes;
    {
        struct stat st;
        uint64_t largest = 0;
        for (const std::string* f : {&o.read_load_file, &o.read_scan_file})
            if (stat(f->c_str(), &st) == 0 && S_ISREG(st.st_mode)) largest = std::max<uint64_t>(largest, (uint64_t)st.st_size);
        const uint64_t share = largest / (uint64_t)o.gpus + (4u << 20);
        if (share < so.chunk_bytes) so.chunk_bytes = ((share >> 20) + 1) << 20;
    }
    ShardedRun run(so);
    if (int rc = run.create()) { fprintf(stderr, "fgpu_create failed (%d): %s\n", rc, run.error().c_str()); return 2; }
    clk.mark("arguments, sizing, contexts");
    std::vector<uint8_t> bloom_bytes(tai / 8);
    std::thread bloom_writer;
    bool bloom_write_failed = false;
    struct JoinWriter {
        std::thread& t;
        ~JoinWriter() { if (t.joinable()) t.join(); }
    } join_writer{bloom_writer};
    auto bloom_file_complete = [&]() -> bool {
        if (bloom_writer.joinable()) bloom_writer.join();
        if (bloom_write_failed) fprintf(stderr, "cannot write %s.bloom\n", o.file_prefix.c_str());
        return !bloom_write_failed;
    };
    fgpu_ctx* ctx = run.first_ctx();      // (CHECK reports through it)
    if (o.from_bloom) {                   // Bloom::load (utils/Bloom.cpp:580-587), the filter on every device
        FILE* f = fopen(o.bloom_input_file.c_str(), "rb");
        if (!f) { fprintf(stderr, "cannot open %s\n", o.bloom_input_file.c_str()); return 2; }
        printf("loading bloom filter from file, nelem %llu \n", (unsigned long long)(tai / 8));
        const size_t got = fread(bloom_bytes.data(), 1, bloom_bytes.size(), f);
        fclose(f);
        if (got != bloom_bytes.size())
            fprintf(stderr, "note: %s holds %llu of the %llu bytes this run's filter has: the rest stays empty (as in the reference)\n",
                    o.bloom_input_file.c_str(), (unsigned long long)got, (unsigned long long)(tai / 8));
        for (int r = 0; r < run.n_ranks(); r++) {
            ctx = run.ctx(r);
            CHECK(fgpu_bloom_upload(ctx, FGPU_BLOO2, bloom_bytes.data(), bloom_bytes.size()));
        }
        printf("bloom loaded\n");
    } else {
        time_t start, stop;
        time(&start);
        printf("Weights before load: %f, %f \n", 0.0f, 0.0f);
        ShardLoadResult lr;
        if (int rc = run.load(o.read_load_file, &lr)) { fprintf(stderr, "load pass failed (%d): %s\n", rc, run.error().c_str()); return 2; }
        clk.mark(lr.fixup ? "pass 1 (shards, fix-up protocol)" : "pass 1 (shards, presence protocol)");
        fprintf(stdout, "\rreads consumed: %lld", (long long)lr.stats.reads_processed);
        printf("\n");
        printf("Weights after load: %f, %f \n", lr.w1, lr.w2);
        printf("Reads processed: %llu\n", (unsigned long long)lr.stats.reads_processed);
        printf("Unambiguous reads: %llu\n", (unsigned long long)lr.stats.unambiguous_reads);
        time(&stop);
        printf("Time to load: %f \n", difftime(stop, start));
        ctx = run.first_ctx();
        CHECK(fgpu_bloom_download(ctx, FGPU_BLOO2, bloom_bytes.data(), bloom_bytes.size()));
        const std::string path = o.file_prefix + ".bloom";       // Bloom::dump, utils/Bloom.cpp:571-578
        FILE* f = fopen(path.c_str(), "wb");
        if (!f) { fprintf(stderr, "cannot write %s\n", path.c_str()); return 2; }
        bloom_writer = std::thread([f, &bloom_bytes, &bloom_write_failed] {
            if (fwrite(bloom_bytes.data(), 1, bloom_bytes.size(), f) != bloom_bytes.size()) bloom_write_failed = true;
            if (fclose(f) != 0) bloom_write_failed = true;
        });
        printf("bloom dumped \n");
        clk.mark("bloom download (dump beside pass 2)");
    }
    PairFilter short_pf, long_pf;        // src/Faucet.cpp:266-283, as in the single-device run
    {
        const uint64_t E = o.estimated_kmers;
        short_pf.create(o.high_cov ? E / 2 : o.mercy ? E / 10 : E / 20, 0.01f);
        if (o.paired_ends) long_pf.create(o.high_cov ? E / 2 : o.mercy ? E / 5 : E / 10, 0.01f);
    }
    if (o.just_load) return bloom_file_complete() ? 0 : 2;
    {
        time_t start, stop;
        time(&start);
        float w2 = 0;
        ctx = run.last_ctx();
        CHECK(fgpu_bloom_weight(ctx, FGPU_BLOO2, &w2));
        printf("Weight before read scan: %f \n", w2);
        run.set_pair_filters(o.no_cleaning ? 0 : short_pf.tai, short_pf.n_hash, o.paired_ends && !o.no_cleaning ? long_pf.tai : 0, long_pf.n_hash);
        ShardScanResult sr;
        if (int rc = run.scan(o.read_scan_file, &sr)) { fprintf(stderr, "junction scan failed (%d): %s\n", rc, run.error().c_str()); return 2; }
        fprintf(stdout, "\rreads scanned: %lld", (long long)sr.stats.reads_processed);
        if (!o.no_cleaning) CHECK(fgpu_scan_short_pairs_download(ctx, short_pf.bits.data(), short_pf.bits.size()));
        if (o.paired_ends && !o.no_cleaning) {
            uint64_t e = 0, ne = 0;      // (the counts are the sum over the shards, sr; the last shard's filter is the run's)
            CHECK(fgpu_scan_long_pairs_download(ctx, long_pf.bits.data(), long_pf.bits.size(), &e, &ne));
        }
        time(&stop);
        clk.mark("pass 2 (shards)");
        if (int rc = write_scan_outputs(o, ctx, sr.stats, sr.empty_count, sr.not_empty_count, difftime(stop, start), short_pf, long_pf, clk)) return rc;
    }
    clk.mark("pair filter weights");
    if (!o.no_cleaning)
        fprintf(stderr, "The contig-graph stage is not part of this build: the load and scan outputs have been written; the reference\n"
                        "continues from the same calls through integration/faucet_binding.cpp (INTEGRATION.md).\n");
    if (!bloom_file_complete()) return 2;
    const int code = o.no_cleaning ? 0 : 3;
    if (!getenv("FGPU_CLI_TIDY")) {      // (as in the single-device run: the driver releases what the process held)
        fflush(stdout);
        fflush(stderr);
        _exit(code);
    }
    return code;
}

int main(int argc, char** argv) {
    PhaseClock clk;
    Options o;
    if (handle_arguments(argc, argv, o) == 1) return 1;
    fgpu_ctx* ctx = nullptr;
    if (o.k < 1 || o.k > 31) { fprintf(stderr, "k must be in 1..31 on this build\n"); return 1; }

    // ---- filter sizing: getBloomFilterFromReads / getBloomFilterFromFile (src/Faucet.cpp:185-219)
    uint64_t tai = 0;
    int32_t n_hash = 0, bits = 0;
    if (o.from_bloom) {
        if (o.two_hash) fgpu_size_two_hash(o.estimated_kmers, o.fp_rate, &bits, &tai, &n_hash);
        else fgpu_size_optimal(o.estimated_kmers, o.fp_rate, &bits, &tai, &n_hash);
        printf("Bits per kmer: %d \n", bits);                       // create_bloom_filter_2_hash / _optimal, utils/Bloom.cpp:204-247
        if (o.two_hash) {
            printf("Estimated items: %llu \n", (unsigned long long)o.estimated_kmers);
            printf("Estimated bloom size: %llu .\n", (unsigned long long)(o.estimated_kmers * (uint64_t)bits));
        }
        printf("BF memory: %f MB\n", (float)((o.estimated_kmers * (uint64_t)bits) / 8ULL / 1024ULL) / 1024);
        printf("Number of hash functions: %d \n", n_hash);
    } else {
        int32_t iters = 0;
        double p1 = fgpu_solve_p1(o.estimated_kmers, o.singletons, o.fp_rate, &iters);
        if (p1 < 0) {   // the reference prints a message and goes on with nonsense sizes; we refuse
            fprintf(stderr, "Signs of f(lower_bound) and f(upper_bound) must be opposites: check -singletons (must be > 0)\n");
            return 1;
        }
        printf("After %d iterations the root is: %g\n", iters, p1);
        printf("p2 is %g p1 estimated as %g\n", (double)o.fp_rate, p1);
        fgpu_size_optimal(o.estimated_kmers, (float)p1, &bits, &tai, &n_hash);
        for (int i = 0; i < 2; i++) {   // printed once per filter, utils/Bloom.cpp:234-243
            printf("Bits per kmer: %d \n", bits);
            printf("BF memory: %f MB\n", (float)((o.estimated_kmers * (uint64_t)bits) / 8ULL / 1024ULL) / 1024);
            printf("Number of hash functions: %d \n", n_hash);
        }
    }

    fgpu_params prm;
    memset(&prm, 0, sizeof(prm));
    prm.k = o.k;
    prm.j = o.j;
    prm.max_spacer_dist = o.max_spacer_dist;
    prm.n_hash = n_hash;
    prm.tai = tai;
    // scanInputRead's lists feed the pair filters and the pair counts, all on the device: the short filter's adds are order-free
    // (fgpu_scan_short_pairs), the long filter's check-then-insert loop is iterated to the sequential result (fgpu_scan_long_pairs); with
    // --no_cleaning only the paired-end loop's two counts are left of them
    const bool record_lists = !o.no_cleaning || o.paired_ends;
    const bool device_short_pairs = !o.no_cleaning;
    if (record_lists) prm.flags |= FGPU_FLAG_RECORD_STOPS;
    prm.flags |= FGPU_FLAG_KEY_ORDER_FROM_START;   // real genomes have repeats: a few launches per window against a first batch walked by cluster
    if (o.mercy) prm.flags |= FGPU_FLAG_MERCY;
    // The text buffers are page-locked, and pinning costs time in proportion (2 x 80 MB: ~40 ms beside the context's creation, on the same
    // driver): input files that are smaller than a chunk -- the reference's own test case is 100 KB -- get buffers of their size.
    o.chunk_bytes = o.chunk_mb << 20;
    {
        uint64_t largest = 0;
        bool all_regular = true;
        for (const std::string* f : {&o.read_load_file, &o.read_scan_file}) {
            struct stat st;
            if (f->empty()) continue;
            if (stat(f->c_str(), &st) == 0 && S_ISREG(st.st_mode)) largest = std::max<uint64_t>(largest, (uint64_t)st.st_size);
            else all_regular = false;
        }
        if (all_regular && largest + (1u << 20) < o.chunk_bytes) o.chunk_bytes = ((largest >> 20) + 1) << 20;
    }
    if (o.gpus > 1 && !o.from_junctions) return main_sharded(o, prm, tai, clk);
    {
        std::thread pin;   // (joined before anything else can fail or read)
        if (!o.batch_reads && !o.from_junctions)
            pin = std::thread([&o] { for (int i = 0; i < 2; i++) pinned_slot(i, kTextPad + o.chunk_bytes); });
        int rc = fgpu_create(&prm, &ctx);
        if (pin.joinable()) pin.join();
        if (rc != FGPU_OK) { fprintf(stderr, "fgpu_create failed (%d): %s\n", rc, fgpu_last_error(nullptr)); return 2; }
    }
    // fgpu_create returns while a thread of the library is still creating the walk's streams and loading code objects: every way out of main
    // goes through fgpu_destroy, which joins it, before the HIP runtime is torn down.  The one exception is the _exit at the very
    // end, after both passes -- the first scan call has joined the thread by then.
    struct CloseContext {
        fgpu_ctx*& c;
        ~CloseContext() { if (c) { fgpu_destroy(c); c = nullptr; } }
    } close_context{ctx};
    clk.mark("arguments, sizing, fgpu_create");
    std::vector<uint8_t> bloom_bytes(tai / 8);
    std::thread bloom_writer;
    bool bloom_write_failed = false;
    struct JoinWriter {      // every way out of main waits for the .bloom file to be complete
        std::thread& t;
        ~JoinWriter() { if (t.joinable()) t.join(); }
    } join_writer{bloom_writer};
    // ... and every SUCCESSFUL way out looks at how that write went (a short write of --just_load_bloom's only output is not exit code 0)
    auto bloom_file_complete = [&]() -> bool {
        if (bloom_writer.joinable()) bloom_writer.join();
        if (bloom_write_failed) fprintf(stderr, "cannot write %s.bloom\n", o.file_prefix.c_str());
        return !bloom_write_failed;
    };

    // ---- pass 1 (load_two_filters, utils/Bloom.cpp:267-350) or -bloom_file (Bloom::load, :580-587)
    if (o.from_bloom) {
        FILE* f = fopen(o.bloom_input_file.c_str(), "rb");
        if (!f) { fprintf(stderr, "cannot open %s\n", o.bloom_input_file.c_str()); return 2; }
        printf("loading bloom filter from file, nelem %llu \n", (unsigned long long)(tai / 8));
        size_t got = fread(bloom_bytes.data(), 1, bloom_bytes.size(), f);
        fclose(f);
        // Bloom::load (utils/Bloom.cpp:580-587) reads as many bytes as the filter it has just sized holds and does not look at fread's
        // count: a longer file gives its head, a shorter one leaves the rest of the (zeroed) filter empty -- e.g. the .bloom of a run that
        // loaded from reads (sized with p1) restarted with the sizes -fp implies.  Same here; a note on stderr is all that is added.
        if (got != bloom_bytes.size())
            fprintf(stderr, "note: %s holds %llu of the %llu bytes this run's filter has: the rest stays empty (as in the reference)\n",
                    o.bloom_input_file.c_str(), (unsigned long long)got, (unsigned long long)(tai / 8));
        CHECK(fgpu_bloom_upload(ctx, FGPU_BLOO2, bloom_bytes.data(), bloom_bytes.size()));
        printf("bloom loaded\n");
    } else {
        BatchSource src(o, o.read_load_file);
        if (!src.is_open()) { fprintf(stderr, "cannot open %s\n", o.read_load_file.c_str()); return 2; }   // the reference silently reads nothing
        time_t start, stop;
        time(&start);
        printf("Weights before load: %f, %f \n", 0.0f, 0.0f);
        CHECK(fgpu_load_begin(ctx, 0));
        uint64_t consumed = 0;
        fgpu_reads r;
        for (int more; (more = src.next(ctx, &r)) != 0;) {
            if (more < 0) { fprintf(stderr, "fgpu_text_split failed (%d): %s\n", -more, fgpu_last_error(ctx)); return 2; }
            const auto t_batch = std::chrono::steady_clock::now();
            CHECK(fgpu_load_batch(ctx, &r));
            clk.batch_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_batch).count();
            consumed += r.n_reads;
            fprintf(stdout, "\rreads consumed: %lld", (long long)consumed);
            fflush(stdout);
        }
        fgpu_load_stats ls;
        const auto t_end = std::chrono::steady_clock::now();
        CHECK(fgpu_load_end(ctx, &ls));
        if (clk.on) fprintf(stderr, "[cli]   %.2f ms in fgpu_load_batch calls, %.2f ms in fgpu_load_end\n", clk.batch_ms,
                            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_end).count());
        clk.mark("pass 1 (read + load)");
        float w1 = 0, w2 = 0;
        CHECK(fgpu_bloom_weight(ctx, FGPU_BLOO1, &w1));
        CHECK(fgpu_bloom_weight(ctx, FGPU_BLOO2, &w2));
        printf("\n");
        printf("Weights after load: %f, %f \n", w1, w2);
        printf("Reads processed: %llu\n", (unsigned long long)ls.reads_processed);
        printf("Unambiguous reads: %llu\n", (unsigned long long)ls.unambiguous_reads);
        time(&stop);
        printf("Time to load: %f \n", difftime(stop, start));
        CHECK(fgpu_bloom_download(ctx, FGPU_BLOO2, bloom_bytes.data(), bloom_bytes.size()));
        const std::string path = o.file_prefix + ".bloom";       // Bloom::dump, utils/Bloom.cpp:571-578
        FILE* f = fopen(path.c_str(), "wb");
        if (!f) { fprintf(stderr, "cannot write %s\n", path.c_str()); return 2; }
        // the bytes are in host memory; writing them out (64 MiB - 1 GiB) runs beside pass 2 and is waited for before the process ends
        bloom_writer = std::thread([f, &bloom_bytes, &bloom_write_failed] {
            if (fwrite(bloom_bytes.data(), 1, bloom_bytes.size(), f) != bloom_bytes.size()) bloom_write_failed = true;
            if (fclose(f) != 0) bloom_write_failed = true;
        });
        printf("bloom dumped \n");
        clk.mark("bloom download (dump beside pass 2)");
    }
    // ---- pair filters (src/Faucet.cpp:266-283): created, and their sizes printed, before --just_load_bloom returns
    PairFilter short_pf, long_pf;
    {
        const uint64_t E = o.estimated_kmers;
        // mercy k-mers lead to more junctions: the reference doubles both filters (src/Faucet.cpp:268-271)
        short_pf.create(o.high_cov ? E / 2 : o.mercy ? E / 10 : E / 20, 0.01f);
        if (o.paired_ends) long_pf.create(o.high_cov ? E / 2 : o.mercy ? E / 5 : E / 10, 0.01f);
    }
    if (o.just_load) return bloom_file_complete() ? 0 : 2;

    // ---- -junctions_file <prefix> (src/Faucet.cpp:104-109,289-293): the scan's three files are reloaded instead of being made.  What the
    // reference does next is its contig graph, which is not part of this build: the files are parsed and checked (sizes as the flags imply),
    // the lines the reference prints after reloading are printed, and the program stops where every run of this build stops (exit code 3).
    if (o.from_junctions) {
        JunctionTable junction_map;
        printf("Reading from Junction file to build junction map.\n");
        const long n_lines = read_junctions(o.junctions_prefix + ".junctions", o.k, junction_map);
        if (n_lines < 0) { fprintf(stderr, "cannot read %s.junctions (missing, or not in the .junctions format for k = %d)\n", o.junctions_prefix.c_str(), o.k); return 2; }
        if (int rc = load_pair_filter(short_pf, o.junctions_prefix + ".short_pair_filter")) return rc;
        if (o.paired_ends)
            if (int rc = load_pair_filter(long_pf, o.junctions_prefix + ".long_pair_filter")) return rc;
        printf("Weight of short pair filter: %f\n", short_pf.weight());
        if (o.paired_ends) printf("Weight of long pair filter: %f\n", long_pf.weight());
        printf("Number of junctions: %llu\n", (unsigned long long)junction_map.size());
        fprintf(stderr, "The contig-graph stage is not part of this build: -bloom_file / -junctions_file inputs have been read and checked; the\n"
                        "reference continues from them.\n");
        return bloom_file_complete() ? 3 : 2;
    }

    // ---- pass 2 (ReadScanner::scanReads, src/ReadScanner.cpp:284-359; printScanSummary :19-27)
    {
        time_t start, stop;
        time(&start);
        float w2 = 0;
        CHECK(fgpu_bloom_weight(ctx, FGPU_BLOO2, &w2));
        printf("Weight before read scan: %f \n", w2);
        uint64_t empty_count = 0, not_empty_count = 0;
        fgpu_scan_stats ss;
        // One scan of the file.  0 = done, 2 = fatal (-1: a status only library versions before the scan journal could return).
        auto scan_once = [&]() -> int {
            BatchSource src(o, o.read_scan_file);
            if (!src.is_open()) { fprintf(stderr, "cannot open %s\n", o.read_scan_file.c_str()); return 2; }
            auto failed = [&](const char* what, int rc) -> int {
                const std::string msg = fgpu_last_error(ctx);
                if (rc == FGPU_ERR_STATE && msg.find("lazy-flag") != std::string::npos) return -1;
                fprintf(stderr, "%s failed (%d): %s\n", what, rc, msg.c_str());
                return 2;
            };
            int rc = device_short_pairs ? fgpu_scan_short_pairs(ctx, short_pf.tai, short_pf.n_hash, 0) : FGPU_OK;
            if (rc != FGPU_OK) return failed("fgpu_scan_short_pairs", rc);
            // scanReads' paired-end loop (src/ReadScanner.cpp:317-343): on the device, from the same lists, in file order
            bool host_long_pairs = false;
            if (o.paired_ends) {
                rc = o.no_cleaning ? fgpu_scan_long_pairs(ctx, 0, 0, FGPU_LONG_PAIRS_COUNT) : fgpu_scan_long_pairs(ctx, long_pf.tai, long_pf.n_hash, FGPU_LONG_PAIRS_FILTER);
                if (rc == FGPU_ERR_NOMEM && !o.no_cleaning) {
                    // the device cannot hold the filter's working state (4 bytes per filter bit; --high_cov sizes the filter at E / 2 x 9 bits,
                    // src/Faucet.cpp:279-280): the loop runs HERE instead, over the lists the device hands out (pair_loop.h) -- the reference
                    // has no such limit, so neither has the command line
                    fprintf(stderr, "note: the long pair filter (%llu bits) does not fit the device's fixed-point form; the paired-end loop runs on the host\n",
                            (unsigned long long)long_pf.tai);
                    host_long_pairs = true;
                    rc = fgpu_scan_long_pairs(ctx, 0, 0, FGPU_LONG_PAIRS_OFF);
                    if (rc == FGPU_OK && device_short_pairs) rc = fgpu_scan_short_pairs(ctx, short_pf.tai, short_pf.n_hash, 1);   // (the lists come to the host)
                }
                if (rc != FGPU_OK) return failed("fgpu_scan_long_pairs", rc);
            }
            faucet_host::HostLongPairs hlp(long_pf.bits.data(), long_pf.tai, long_pf.n_hash, o.k, true);
            std::vector<fgpu_stop> stops;
            std::vector<uint64_t> batch_reads;
            auto take_lists = [&](bool all) -> int {     // one batch's lists (after a batch call: the batch before it), or all that are left
                for (;;) {
                    uint64_t n_stops = 0;
                    int64_t seq = -1;
                    int trc = fgpu_scan_take_stops(ctx, stops.data(), stops.size(), &n_stops, &seq);
                    if (trc == FGPU_ERR_CAPACITY) { stops.resize((size_t)(n_stops + n_stops / 4 + 16)); continue; }
                    if (trc != FGPU_OK) return failed("fgpu_scan_take_stops", trc);
                    if (seq < 0) return 0;
                    hlp.batch(stops.data(), n_stops, batch_reads[(size_t)seq]);
                    if (!all) return 0;
                }
            };
            rc = fgpu_scan_begin(ctx);
            if (rc != FGPU_OK) return failed("fgpu_scan_begin", rc);
            uint64_t scanned = 0;
            fgpu_reads r;
            for (int more; (more = src.next(ctx, &r)) != 0;) {
                if (more < 0) return failed("fgpu_text_split", -more);
                const auto t_scan = std::chrono::steady_clock::now();
                if ((rc = fgpu_scan_batch(ctx, &r)) != FGPU_OK) return failed("fgpu_scan_batch", rc);
                clk.scan_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_scan).count();
                if (host_long_pairs) {
                    batch_reads.push_back(r.n_reads);
                    if (batch_reads.size() > 1) { const int trc = take_lists(false); if (trc) return trc; }
                }
                scanned += r.n_reads;
                fprintf(stdout, "\rreads scanned: %lld", (long long)scanned);
                fflush(stdout);
            }
            const auto t_end = std::chrono::steady_clock::now();
            if ((rc = fgpu_scan_end(ctx, &ss)) != FGPU_OK) return failed("fgpu_scan_end", rc);
            clk.take_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_end).count();
            if (device_short_pairs && (rc = fgpu_scan_short_pairs_download(ctx, short_pf.bits.data(), short_pf.bits.size())) != FGPU_OK)
                return failed("fgpu_scan_short_pairs_download", rc);
            if (host_long_pairs) {
                const int trc = take_lists(true);
                if (trc) return trc;
                empty_count = hlp.empty_count;
                not_empty_count = hlp.not_empty_count;
            } else
            if (o.paired_ends) {
                rc = o.no_cleaning ? fgpu_scan_long_pairs_download(ctx, nullptr, 0, &empty_count, &not_empty_count)
                                   : fgpu_scan_long_pairs_download(ctx, long_pf.bits.data(), long_pf.bits.size(), &empty_count, &not_empty_count);
                if (rc != FGPU_OK) return failed("fgpu_scan_long_pairs_download", rc);
            }
            return 0;
        };
        // (a preview of the junction walk that the library cannot repair is absorbed inside the library: it keeps the scan's batches in HBM and
        // scans them again by itself -- nothing is read twice here, which is what lets both inputs be pipes)
        const int src_rc = scan_once();
        if (src_rc == -1) { fprintf(stderr, "scan failed: %s\n", fgpu_last_error(ctx)); return 2; }
        if (src_rc) return src_rc;
        time(&stop);
        if (clk.on) fprintf(stderr, "[cli]   %.2f ms in fgpu_scan_batch calls, %.2f ms in fgpu_scan_end (the last walks, the last lists, the pair filters' last batches)\n",
                            clk.scan_ms, clk.take_ms);
        if (clk.on) {
            uint64_t waits = 0;
            double wait_ms = 0;
            if (fgpu_diag_host_waits(ctx, &waits, &wait_ms) == FGPU_OK)
                fprintf(stderr, "[cli]   pass 2: the host waited for the device %llu times, %.2f ms in all\n", (unsigned long long)waits, wait_ms);
        }
        if (clk.on && o.paired_ends && !o.no_cleaning) {
            uint64_t d[6] = {0, 0, 0, 0, 0, 0};
            if (fgpu_diag_long_pairs(ctx, d) == FGPU_OK)
                fprintf(stderr, "[cli]   long pair filter on the device: %llu first-end k-mers checked, %llu paired by the filter as their batch found it, %llu inserted; "
                                "%llu evaluation rounds over %llu batches (at most %llu in one)\n",
                        (unsigned long long)d[0], (unsigned long long)d[1], (unsigned long long)d[2], (unsigned long long)d[3], (unsigned long long)d[5], (unsigned long long)d[4]);
        }
        if (clk.on) {
            uint64_t d[6] = {0, 0, 0, 0, 0, 0};
            if (fgpu_diag_ovw(ctx, d) == FGPU_OK && (d[2] || d[3] || d[5]))
                fprintf(stderr, "[cli]   large clusters, walked optimistically: %llu pieces, %llu rounds over %llu windows (%llu piece-rounds kept their log); "
                                "%llu windows left to the key-ordered walk, %llu to the walk by cluster\n",
                        (unsigned long long)d[0], (unsigned long long)d[1], (unsigned long long)d[2], (unsigned long long)d[4], (unsigned long long)d[3],
                        (unsigned long long)d[5]);
        }
        clk.mark("pass 2 (read + scan)");
        if (int rc = write_scan_outputs(o, ctx, ss, empty_count, not_empty_count, difftime(stop, start), short_pf, long_pf, clk)) return rc;
    }
    clk.mark("pair filter weights");
    if (!o.no_cleaning)
        fprintf(stderr, "The contig-graph stage is not part of this build: the load and scan outputs have been written; the reference\n"
                        "continues from the same calls through integration/faucet_binding.cpp (INTEGRATION.md).\n");
    if (!bloom_file_complete()) return 2;                  // (_exit below skips destructors)
    const int code = o.no_cleaning ? 0 : 3;
    // Every output file is closed and both passes have ended with a synchronised device.  Returning several hundred HBM buffers one by
    // one, unpinning the text buffers and unloading the HIP runtime took 0.15 s of a 0.9 s run and changes nothing the caller can see:
    // the driver releases what the process held.  FGPU_CLI_TIDY=1 takes the long way (profilers that write their output at exit need it).
    if (!getenv("FGPU_CLI_TIDY")) {
        fflush(stdout);
        fflush(stderr);
        _exit(code);
    }
    fgpu_destroy(ctx);
    ctx = nullptr;
    clk.mark("fgpu_destroy");
    return code;
}
