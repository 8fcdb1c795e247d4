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
// leaves there: the JunctionMap container that fixes the dump order, and the two pair filters, which are fed from
// scanInputRead's per-read lists (fgpu_scan_take_stops) exactly as ReadScanner does it.
#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <sched.h>
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
#include <deque>
#include <fstream>
#include <mutex>
#include <new>
#include <thread>
#include <string>
#include <unordered_map>
#include <vector>

#include "faucet_gpu.h"
#include "junction_order.h"

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

// The two text buffers of the device-split input are pinned (the copy to the device runs at link speed) and belong to the process, not to
// one pass: pinning and unpinning 160 MB costs 30 ms each way, which was a fifth of a pass over a 1 GB file.  main() pins them on a helper
// thread while the context is being created; they are returned when the process ends.
static char* pinned_slot(int i, uint64_t bytes) {
    static char* base[2] = {nullptr, nullptr};
    static uint64_t have[2] = {0, 0};
    static std::mutex m;
    std::lock_guard<std::mutex> g(m);
    if (have[i] < bytes) {
        if (base[i]) fgpu_host_free(base[i]);
        base[i] = (char*)fgpu_host_alloc(bytes);
        if (!base[i]) base[i] = (char*)malloc(bytes);
        have[i] = base[i] ? bytes : 0;
    }
    return base[i];
}
static const uint64_t kTextPad = 16u << 20;   // room in front of a chunk for the unconsumed tail of the previous one

// The same loop with the records split on the device (fgpu_text_split): the host only moves file text, `chunk` bytes at a
// time, and carries the unconsumed tail (an incomplete record) over to the next call.  Works on non-seekable input.
class TextSource {
public:
    TextSource(const std::string& path, bool fastq, uint64_t chunk) : fd_(open(path.c_str(), O_RDONLY)), fastq_(fastq), chunk_(chunk) {
        if (fd_ < 0) return;
        struct stat st;
        regular_ = fstat(fd_, &st) == 0 && S_ISREG(st.st_mode);
        for (int i = 0; i < 2; i++) slot_[i].base = pinned_slot(i, kPad + chunk_);
        reader_ = std::thread(&TextSource::read_ahead, this);
    }
    ~TextSource() {
        if (fd_ < 0) return;
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
        }
        cv_.notify_all();
        reader_.join();
        close(fd_);
        if (getenv("FGPU_CLI_TIMES"))
            fprintf(stderr, "[cli]   text source: %.2f ms waiting for the reader, %.2f ms in fgpu_text_split, %.2f ms reading (reader thread)\n",
                    wait_ms_, split_ms_, read_ms_);
    }
    bool is_open() const { return fd_ >= 0 && slot_[0].base && slot_[1].base; }
    // 1 = a batch (device pointers, valid until the next call), 0 = input exhausted, < 0 = -status of a failed call
    int next(fgpu_ctx* ctx, fgpu_reads* out) {
        for (;;) {
            if (finished_) return 0;
            Slot& sl = slot_[cur_];
            const auto t_wait = std::chrono::steady_clock::now();
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return sl.full; });
            }
            const auto t_split = std::chrono::steady_clock::now();
            wait_ms_ += std::chrono::duration<double, std::milli>(t_split - t_wait).count();
            // the unconsumed tail of the previous chunk sits right in front of this chunk's text
            char* text = sl.base + kPad - tail_;
            const uint64_t n = tail_ + sl.got;
            uint64_t used = 0;
            const int rc = fgpu_text_split(ctx, text, n, 0, fastq_ ? 1 : 0, sl.eof ? 1 : 0, out, &used);
            split_ms_ += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_split).count();
            if (rc != FGPU_OK) return -rc;
            const uint64_t left = n - used;
            if (sl.eof) {
                finished_ = true;
            } else {
                if (left > kPad) return -FGPU_ERR_CAPACITY;   // a single line of more than 16 MB: use -batch_reads (host getline)
                memcpy(slot_[cur_ ^ 1].base + kPad - left, text + used, left);
                tail_ = left;
                {
                    std::lock_guard<std::mutex> g(m_);
                    sl.full = false;
                }
                cv_.notify_all();
                cur_ ^= 1;
            }
            if (out->n_reads) return 1;
            // no complete record inside a whole chunk: its text has become the tail, read on
        }
    }
private:
    static constexpr uint64_t kPad = kTextPad;
    struct Slot {
        char* base = nullptr;
        size_t got = 0;
        bool eof = false, full = false;
    };
    // The first batches of both passes are the dear ones per read (empty carry: every occurrence goes through the resolve kernel;
    // empty junction map: every junction test is evaluated), so the input starts with smaller chunks: 1/4, 1/4, 1/2 of a chunk,
    // full chunks from then on.  A function of the chunk index only: both passes cut the same file into the same batches, which
    // is what lets the scan reuse the planes the load kept.
    size_t chunk_bytes(uint64_t i) const {
        const uint64_t w = i < 2 ? chunk_ / 4 : i == 2 ? chunk_ / 2 : chunk_;
        return (size_t)std::max<uint64_t>(w, std::min<uint64_t>(chunk_, 64u << 10));
    }
    void read_ahead() {   // reader thread: keeps the other slot filled while the device works on the current one
        for (int i = 0;; i ^= 1) {
            Slot& sl = slot_[i];
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return stop_ || !sl.full; });
                if (stop_) return;
            }
            const size_t want = chunk_bytes(n_chunks_++);
            const auto t_read = std::chrono::steady_clock::now();
            const size_t got = fill(sl.base + kPad, want);
            read_ms_ += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_read).count();
            {
                std::lock_guard<std::mutex> g(m_);
                sl.got = got;
                sl.eof = got < want;
                sl.full = true;
            }
            cv_.notify_all();
            if (got < want) return;
        }
    }
    // `want` bytes from the input, fewer only at its end.  A regular file is read by kReaders threads at once (one thread copies out of
    // the page cache at 7-8 GB/s, which was 2.4 times the time the device needs for the same text); anything else (a pipe, a process
    // substitution) is read in order by this thread alone.
    static size_t read_fully(int fd, char* dst, size_t want, off_t at, bool positioned) {
        size_t got = 0;
        while (got < want) {
            const ssize_t r = positioned ? pread(fd, dst + got, want - got, at + (off_t)got) : read(fd, dst + got, want - got);
            if (r < 0 && errno == EINTR) continue;
            if (r <= 0) break;
            got += (size_t)r;
        }
        return got;
    }
    size_t fill(char* dst, size_t want) {
        if (!regular_ || want < (8u << 20)) {
            const size_t got = read_fully(fd_, dst, want, (off_t)offset_, regular_);
            offset_ += got;
            return got;
        }
        const size_t part = ((want + kReaders - 1) / kReaders + 4095) & ~(size_t)4095;
        size_t got_part[kReaders] = {0};
        std::thread helpers[kReaders];
        for (unsigned t = 1; t < kReaders; t++)
            if ((size_t)t * part < want)
                helpers[t] = std::thread([&, t] { got_part[t] = read_fully(fd_, dst + t * part, std::min(part, want - t * part), (off_t)(offset_ + t * part), true); });
        got_part[0] = read_fully(fd_, dst, std::min(part, want), (off_t)offset_, true);
        size_t got = 0;
        bool short_part = false;
        for (unsigned t = 0; t < kReaders; t++) {
            if (helpers[t].joinable()) helpers[t].join();
            if (!short_part) got += got_part[t];
            if ((size_t)t * part < want && got_part[t] < std::min(part, want - t * part)) short_part = true;   // the file ends inside this part
        }
        offset_ += got;
        return got;
    }
    static constexpr unsigned kReaders = 4;
    double wait_ms_ = 0, split_ms_ = 0, read_ms_ = 0;
    int fd_;
    bool regular_ = false;
    uint64_t offset_ = 0;
    bool fastq_;
    uint64_t chunk_;
    Slot slot_[2];
    int cur_ = 0;
    uint64_t tail_ = 0, n_chunks_ = 0;
    bool finished_ = false, stop_ = false;
    std::thread reader_;
    std::mutex m_;
    std::condition_variable cv_;
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

// ---- the pair filters (host side, as in the reference) ---------------------------------------------------------
uint64_t revcomp(uint64_t x, int k) {   // utils/Kmer.cpp:238-252: complement every 2-bit code (x ^ 2), reverse their order
    x ^= 0xAAAAAAAAAAAAAAAAULL;
    x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
    return __builtin_bswap64(x) >> (64 - 2 * k);
}
uint64_t canonical(uint64_t x, int k) { const uint64_t r = revcomp(x, k); return x < r ? x : r; }   // utils/Kmer.cpp:531-533

uint64_t old_hash(uint64_t key, uint64_t seed) {   // Bloom::oldHash, utils/Bloom.h:134-145
    uint64_t h = seed;
    h ^= (h << 7) ^ key * (h >> 3) ^ (~((h << 11) + (key ^ (h >> 5))));
    h = (~h) + (h << 21);
    h = h ^ (h >> 24);
    h = (h + (h << 3)) + (h << 8);
    h = h ^ (h >> 14);
    h = (h + (h << 2)) + (h << 4);
    h = h ^ (h >> 28);
    h = h + (h << 31);
    return h;
}
const uint64_t kSeed0 = 0xffaa54ffe6e6e6e7ULL, kSeed1 = 0x1140aada557088a4ULL;   // seed_tab[0..1], utils/Bloom.h:56-68 with user_seed 0

// The pair filters' bit arrays on 2 MiB pages where the kernel gives them (transparent huge pages, madvise): the long-pair loop probes
// tens of MB at random, and with 4 KiB pages most of its probes missed the TLB as well (-17 % per check in a stand-alone copy of the loop).
template <class T>
struct HugePageAlloc {
    typedef T value_type;
    HugePageAlloc() {}
    template <class U> HugePageAlloc(const HugePageAlloc<U>&) {}
    T* allocate(size_t n) {
        const size_t kHuge = (size_t)1 << 21;
        const size_t bytes = (n * sizeof(T) + kHuge - 1) / kHuge * kHuge;
        void* p = nullptr;
        if (posix_memalign(&p, kHuge, bytes)) throw std::bad_alloc();
        (void)madvise(p, bytes, MADV_HUGEPAGE);      // advice only: without it the array lives on ordinary pages
        return (T*)p;
    }
    void deallocate(T* p, size_t) { free(p); }
    template <class U> bool operator==(const HugePageAlloc<U>&) const { return true; }
    template <class U> bool operator!=(const HugePageAlloc<U>&) const { return false; }
};

struct PairFilter {   // a Bloom used through addPair / containsPair only (utils/Bloom.cpp:127-154, Bloom.h:217-258)
    uint64_t tai = 0;
    int n_hash = 0;
    std::vector<uint8_t, HugePageAlloc<uint8_t> > bits;
    void create(uint64_t elements, float fp) {   // create_bloom_filter_optimal, utils/Bloom.cpp:229-247
        int32_t bpk = 0, nh = 0;
        fgpu_size_optimal(elements, fp, &bpk, &tai, &nh);
        n_hash = nh;
        printf("Bits per kmer: %d \n", bpk);
        printf("BF memory: %f MB\n", (float)((elements * (uint64_t)bpk) / 8ULL / 1024ULL) / 1024);
        printf("Number of hash functions: %d \n", n_hash);
        bits.assign(tai / 8, 0);
    }
    // e1, e2: the canonical forms of the two k-mers of a JuncPair
    void add_canon(uint64_t e1, uint64_t e2) {
        uint64_t h0 = old_hash(std::min(e1, e2), kSeed0) & (tai - 1);
        const uint64_t h1 = old_hash(std::max(e1, e2), kSeed1) & (tai - 1);
        for (int i = 0; i < n_hash; i++) { bits[h0 >> 3] |= (uint8_t)(1u << (h0 & 7)); h0 = (h0 + h1) & (tai - 1); }
    }
    bool contains_canon(uint64_t e1, uint64_t e2) const {
        uint64_t h0 = old_hash(std::min(e1, e2), kSeed0) & (tai - 1);
        if (!(bits[h0 >> 3] & (1u << (h0 & 7)))) return false;      // most pairs are new: one hash, one probe
        const uint64_t h1 = old_hash(std::max(e1, e2), kSeed1) & (tai - 1);
        for (int i = 1; i < n_hash; i++) { h0 = (h0 + h1) & (tai - 1); if (!(bits[h0 >> 3] & (1u << (h0 & 7)))) return false; }
        return true;
    }
    void add_pair(uint64_t k1, uint64_t k2, int k) { add_canon(canonical(k1, k), canonical(k2, k)); }
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
        fwrite(bits.data(), 1, bits.size(), f);
        fclose(f);
        return 0;
    }
};

// What ReadScanner does with scanInputRead's lists: the short-pair rules at the end of scan_forward
// (src/ReadScanner.cpp:208-225) per valid piece, and the paired-end loop of scanReads (:317-343) per read pair.
struct PairLogic {
    int k = 0;
    bool paired_ends = false, no_cleaning = false;
    PairFilter* short_pf = nullptr;
    PairFilter* long_pf = nullptr;
    bool first_end = true;
    int empty_count = 0, not_empty_count = 0;
    double prepare_ms = 0;                // FGPU_CLI_TIMES: of the worker's time, what `prepare` took

    void piece(const fgpu_stop* s, size_t n) {   // one scan_forward call
        if (no_cleaning || !short_pf) return;
        if (n == 2) {
            bool have_first_back = false, have_last_fwd = false;
            uint32_t rev_pos = 0, for_pos = 0;
            uint64_t first_back = 0, last_fwd = 0;
            for (size_t i = 0; i < n; i++) {
                const uint32_t pos = s[i].info & FGPU_STOP_POS_MASK;
                if (s[i].info & FGPU_STOP_FAKE) continue;
                if (!(s[i].info & FGPU_STOP_FORWARD)) {
                    if (!have_first_back) { have_first_back = true; first_back = s[i].ext; rev_pos = pos; }
                } else {
                    if (!have_last_fwd) { have_last_fwd = true; for_pos = pos; }
                    last_fwd = s[i].ext;
                }
            }
            if (have_first_back && have_last_fwd && !(rev_pos > for_pos)) short_pf->add_pair(first_back, last_fwd, k);
            if ((have_first_back && !have_last_fwd) || (!have_first_back && have_last_fwd)) short_pf->add_pair(s[0].ext, s[1].ext, k);
        } else if (n > 2) {
            for (size_t i = 0; i + 2 < n; i++) short_pf->add_pair(s[i].ext, s[i + 2].ext, k);
        }
    }
    // One end's list as the long-pair loop needs it: the canonical k-mers (all a JuncPair is hashed by) and their two hashes, masked.
    // containsPair / addPair hash the SMALLER k-mer of a pair with one seed and the larger with the other (PairFilter::contains_canon /
    // add_canon), so the |end 1| x |end 2| checks of a read pair need one pair of hashes per list entry, not one per check.
    struct EndList {                       // a view into the batch's arrays (prepare); end 1 is copied when its pair straddles two batches
        const uint64_t *canon = nullptr, *h0 = nullptr, *h1 = nullptr;
        size_t n = 0;
        bool empty() const { return n == 0; }
        size_t size() const { return n; }
    };
    EndList end1, end2;
    std::vector<uint64_t> kept_c, kept_0, kept_1;
    // The long-pair loop in two phases per batch of lists (speculate, then settle in order).  Phase 1, helper threads, the filter only READ:
    // every first-end k-mer of every read pair of the batch is checked against the filter as it stands when the batch begins.  Bits are only
    // ever set, so "paired" found there is final -- that k-mer does nothing when its turn comes.  For the others phase 1 leaves, per mate, the
    // position of the FIRST bit it found missing.  Phase 2, this thread, file order: a k-mer that was not paired looks at those positions
    // again -- one probe per mate; a bit that is still clear means the pair is still absent, exactly -- does a full check only where the bit
    // has been set since (by this batch's earlier inserts), and inserts if nothing turned up.  Same answers and same inserts in the same
    // order as the plain loop (long_pairs_plain), which still takes pairs that straddle two batches and filters of more than 2^32 bits.
    std::vector<uint8_t> spec_found;       // per stop of the batch: (first-end entries) 1 = paired when the batch began
    std::vector<uint32_t> spec_miss;
    struct SpecPair { size_t a1, n1, a2, n2; uint64_t base; };   // base: where the pair's n1 x n2 positions start in spec_miss
    std::vector<SpecPair> spec_pairs;
    size_t spec_next = 0;                  // phase 2: the next speculated pair in file order
    std::vector<size_t> first_stop;        // of every read of the batch (+ one past the end)
    // canonical form and hashes of every stop of a batch, made by helper threads before the sequential loop (prepare): per stop a reverse
    // complement and two oldHash -- 23 M stops on BASELINE config 3's shape, half of what the worker thread used to spend
    std::vector<uint64_t> bc, b0, b1;

    void read(const fgpu_stop* s, size_t n, size_t first) {   // one iteration of the loop in scanReads; `first`: index of s[0] in the batch
        EndList& e = first_end ? end1 : end2;
        e.n = 0;
        size_t a = 0;
        while (short_pf && !no_cleaning && a < n) {
            size_t b = a + 1;
            while (b < n && !(s[b].info & FGPU_STOP_FIRST)) b++;
            piece(s + a, b - a);
            a = b;
        }
        if (paired_ends) {
            e.canon = bc.data() + first;
            if (!b0.empty()) { e.h0 = b0.data() + first; e.h1 = b1.data() + first; }   // (no hashes without cleaning)
            e.n = n;
        }
        if (paired_ends && !first_end) {
            if (!end1.empty() && !end2.empty()) {
                not_empty_count++;
                if (!no_cleaning) long_pairs();
            } else {
                empty_count++;
            }
        }
        first_end = !first_end;
    }
    // The check-then-insert loop over the two ends' lists (src/ReadScanner.cpp:317-343): for every k-mer of the first end, is it paired
    // with ANY k-mer of the second end already?  If not, it is paired with the second end's first one.
#ifdef FGPU_CLI_PROFILE
    unsigned long long tk_long = 0, tk_ins = 0, tk_read = 0, n_chain = 0, n_first_fail = 0, n_hit = 0;
#define TK() __builtin_ia32_rdtsc()
#else
#define TK() 0ULL
#endif
    void long_pairs() {
#ifdef FGPU_CLI_PROFILE
        const unsigned long long tq = TK();
        long_pairs_inner();
        tk_all += TK() - tq;
    }
    unsigned long long tk_all = 0;
    void long_pairs_inner() {
#endif
        if (spec_next < spec_pairs.size() && end1.canon == bc.data() + spec_pairs[spec_next].a1) {   // the pair phase 1 looked at
            long_pairs_settle(spec_pairs[spec_next].base);
            spec_next++;
            return;
        }
        long_pairs_plain();
    }
    // phase 2 of the speculated form (see spec_found)
    void long_pairs_settle(uint64_t base) {
        const uint64_t mask = long_pf->tai - 1;
        const int nh = long_pf->n_hash;
        uint8_t* const bits = long_pf->bits.data();
        const size_t n1 = end1.size(), n2 = end2.size();
        const size_t at1 = (size_t)(end1.canon - bc.data());
        for (size_t i = 0; i < n1; i++) {
            if (spec_found[at1 + i]) continue;                               // paired when the batch began: paired now
            const uint32_t* const miss = spec_miss.data() + base + i * n2;
            const uint64_t p1 = end1.canon[i];
            bool paired = false;
            for (size_t j = 0; j < n2 && !paired; j++) {
                const uint32_t pos = miss[j];
                if (!((bits[pos >> 3] >> (pos & 7)) & 1u)) continue;          // the bit that was missing still is: the pair is still absent
                const bool first_is_smaller = p1 <= end2.canon[j];
                uint64_t h0 = first_is_smaller ? end1.h0[i] : end2.h0[j];
                const uint64_t h1 = first_is_smaller ? end2.h1[j] : end1.h1[i];
                bool all = true;
                for (int t = 0; t < nh && all; t++) { all = ((bits[h0 >> 3] >> (h0 & 7)) & 1u) != 0; h0 = (h0 + h1) & mask; }
                paired = all;
            }
            if (!paired) {                                                    // addPair(pair1, back2.front())
                const bool first_is_smaller = p1 <= end2.canon[0];
                uint64_t h0 = first_is_smaller ? end1.h0[i] : end2.h0[0];
                const uint64_t h1 = first_is_smaller ? end2.h1[0] : end1.h1[i];
                for (int t = 0; t < nh; t++) { bits[h0 >> 3] |= (uint8_t)(1u << (h0 & 7)); h0 = (h0 + h1) & mask; }
            }
        }
    }
    // phase 1 for the read pairs [from, to) of spec_pairs: read-only on the filter, writes only its own entries of spec_found / spec_miss
    void speculate(size_t from, size_t to) {
        const uint64_t mask = long_pf->tai - 1;
        const int nh = long_pf->n_hash;
        const uint8_t* const bits = long_pf->bits.data();
        for (size_t p = from; p < to; p++) {
            const SpecPair sp = spec_pairs[p];
            for (size_t i = 0; i < sp.n1; i++) {
                const uint64_t p1 = bc[sp.a1 + i];
                uint32_t* const miss = spec_miss.data() + sp.base + i * sp.n2;
                bool found = false;
                for (size_t j = 0; j < sp.n2 && !found; j++) {
                    const bool first_is_smaller = p1 <= bc[sp.a2 + j];
                    uint64_t h0 = first_is_smaller ? b0[sp.a1 + i] : b0[sp.a2 + j];
                    const uint64_t h1 = first_is_smaller ? b1[sp.a2 + j] : b1[sp.a1 + i];
                    int t = 0;
                    for (; t < nh; t++) {
                        if (!((bits[h0 >> 3] >> (h0 & 7)) & 1u)) break;
                        h0 = (h0 + h1) & mask;
                    }
                    if (t == nh) found = true;
                    else miss[j] = (uint32_t)h0;
                }
                spec_found[sp.a1 + i] = found ? 1 : 0;
            }
        }
    }
    void long_pairs_plain() {
        const unsigned long long tk0 = TK();
        const uint64_t mask = long_pf->tai - 1;
        const int nh = long_pf->n_hash;
        uint8_t* const bits = long_pf->bits.data();
        const size_t n1 = end1.size(), n2 = end2.size();
        for (size_t i = 0; i < n1; i++) {
            const uint64_t p1 = end1.canon[i];
            bool paired = false;
            // The first probe of a check goes to the SMALLER k-mer's first bit -- set as soon as any pair with that k-mer has been inserted,
            // i.e. nearly always inside a repeat (89 % of 4.2e7 checks on BASELINE config 3's shape): it is the second and third probe that
            // tell pairs apart, and a branch per probe was a misprediction per probe.  The first three probes are taken without branching.
            const uint64_t a0 = end1.h0[i], a1 = end1.h1[i];
            const int nb = nh < 3 ? nh : 3;
            for (size_t j = 0; j < n2 && !paired; j++) {
                const bool first_is_smaller = p1 <= end2.canon[j];           // std::min / std::max of the two canonical k-mers
                uint64_t h0 = first_is_smaller ? a0 : end2.h0[j];
                const uint64_t h1 = first_is_smaller ? end2.h1[j] : a1;
                unsigned all = (bits[h0 >> 3] >> (h0 & 7)) & 1u;
                for (int t = 1; t < nb; t++) { h0 = (h0 + h1) & mask; all &= (bits[h0 >> 3] >> (h0 & 7)) & 1u; }
#ifdef FGPU_CLI_PROFILE
                n_first_fail += !all;
                n_chain += all;
#endif
                if (!all) continue;
                for (int t = nb; t < nh && all; t++) { h0 = (h0 + h1) & mask; all = (bits[h0 >> 3] >> (h0 & 7)) & 1u; }
                paired = all != 0;
            }
#ifdef FGPU_CLI_PROFILE
            n_hit += paired;
#endif
            if (!paired) {                                                    // addPair(pair1, back2.front())
                const unsigned long long tk1 = TK();
                const bool first_is_smaller = p1 <= end2.canon[0];
                uint64_t h0 = first_is_smaller ? end1.h0[i] : end2.h0[0];
                const uint64_t h1 = first_is_smaller ? end2.h1[0] : end1.h1[i];
                for (int t = 0; t < nh; t++) { bits[h0 >> 3] |= (uint8_t)(1u << (h0 & 7)); h0 = (h0 + h1) & mask; }
#ifdef FGPU_CLI_PROFILE
                tk_ins += TK() - tk1;
#else
                (void)tk1;
#endif
            }
        }
#ifdef FGPU_CLI_PROFILE
        tk_long += TK() - tk0;
#else
        (void)tk0;
#endif
    }
    // canonical forms, hashes and -- the same pass over the stops -- where every read's stops begin (first_stop)
    void prepare(const fgpu_stop* stops, size_t n, uint64_t n_reads) {
        bc.resize(n);
        const bool hashes = !no_cleaning && long_pf;
        if (hashes) { b0.resize(n); b1.resize(n); }
        const uint64_t mask = hashes ? long_pf->tai - 1 : 0;
        first_stop.resize(n_reads + 1);
        auto part = [&](size_t from, size_t to) {
            for (size_t i = from; i < to; i++) {
                const uint64_t c = canonical(stops[i].ext, k);
                bc[i] = c;
                if (hashes) { b0[i] = old_hash(c, kSeed0) & mask; b1[i] = old_hash(c, kSeed1) & mask; }
                // stop i is the first one of its read (and of the reads without stops before it)
                const uint64_t r_here = stops[i].read, r_before = i ? stops[i - 1].read + 1 : 0;
                for (uint64_t r = r_before; r <= r_here && r <= n_reads; r++) first_stop[r] = i;
            }
        };
        const size_t kPerThread = 1 << 16;
        const size_t n_threads = std::min<size_t>(4, n / kPerThread);
        if (n_threads < 2) {
            part(0, n);
        } else {
            std::vector<std::thread> th;
            for (size_t t = 1; t < n_threads; t++) th.emplace_back(part, n * t / n_threads, n * (t + 1) / n_threads);
            part(0, n / n_threads);
            for (std::thread& t : th) t.join();
        }
        for (uint64_t r = n ? stops[n - 1].read + 1 : 0; r <= n_reads; r++) first_stop[r] = n;   // the reads after the last stop's, and the end
    }
    double sec_ms[6] = {0, 0, 0, 0, 0, 0};   // -DFGPU_CLI_PROFILE: prepare, read offsets, pair list, phase 1, the loop over reads, of which long_pairs
    void batch(const fgpu_stop* stops, size_t n_stops, uint64_t n_reads) {   // reads of a batch, in file order
        const auto t0 = std::chrono::steady_clock::now();
        auto lap = [&](int which, std::chrono::steady_clock::time_point& from) {
            const auto now = std::chrono::steady_clock::now();
            sec_ms[which] += std::chrono::duration<double, std::milli>(now - from).count();
            from = now;
        };
        auto tl = t0;
        if (paired_ends) {
            prepare(stops, n_stops, n_reads);
        } else {
            first_stop.resize(n_reads + 1);
            size_t a = 0;
            for (uint64_t r = 0; r < n_reads; r++) {
                first_stop[r] = a;
                while (a < n_stops && stops[a].read == r) a++;
            }
            first_stop[n_reads] = a;
        }
        lap(0, tl);
        // the read pairs that lie inside this batch, and room for their mates' positions (at most 2^26 per batch: what is beyond takes the plain loop)
        spec_pairs.clear();
        spec_next = 0;
        static const bool no_spec = getenv("FGPU_CLI_NO_SPEC") != nullptr;     // (measurement: the plain loop for every pair)
        if (!no_spec && paired_ends && !no_cleaning && long_pf && long_pf->tai <= (1ULL << 32) && n_stops) {
            spec_found.resize(n_stops);
            uint64_t room = 0;
            for (uint64_t r = first_end ? 0 : 1; r + 1 < n_reads; r += 2) {   // (first_end: read 0 opens a pair; else it closes one that began in the last batch)
                const SpecPair sp = {first_stop[r], first_stop[r + 1] - first_stop[r], first_stop[r + 1], first_stop[r + 2] - first_stop[r + 1], room};
                if (!sp.n1 || !sp.n2 || room + (uint64_t)sp.n1 * sp.n2 > (1ULL << 26)) continue;
                room += (uint64_t)sp.n1 * sp.n2;
                spec_pairs.push_back(sp);
            }
            spec_miss.resize((size_t)room);
            lap(2, tl);
            const size_t n_threads = std::min<size_t>(6, spec_pairs.size() / 4096);
            if (n_threads < 2) {
                speculate(0, spec_pairs.size());
            } else {
                std::vector<std::thread> th;
                for (size_t t = 1; t < n_threads; t++)
                    th.emplace_back(&PairLogic::speculate, this, spec_pairs.size() * t / n_threads, spec_pairs.size() * (t + 1) / n_threads);
                speculate(0, spec_pairs.size() / n_threads);
                for (std::thread& t : th) t.join();
            }
        }
        prepare_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        lap(3, tl);
        for (uint64_t r = 0; r < n_reads; r++) read(stops + first_stop[r], first_stop[r + 1] - first_stop[r], first_stop[r]);
        lap(4, tl);
        if (paired_ends && !first_end) {          // a first end waits for its mate in the next batch: its list leaves the batch's arrays
            kept_c.assign(end1.canon, end1.canon + end1.n);
            end1.canon = kept_c.data();
            if (!no_cleaning && long_pf) {
                kept_0.assign(end1.h0, end1.h0 + end1.n);
                kept_1.assign(end1.h1, end1.h1 + end1.n);
                end1.h0 = kept_0.data();
                end1.h1 = kept_1.data();
            }
        }
    }
};

// The junction container of the reference, for the paths that need a real one: -junctions_file (a later line with the same k-mer replaces the
// earlier one) and the dump order on a standard library whose container DumpOrder does not replay (junction_order.h).
typedef std::unordered_map<uint64_t, Junction> JunctionTable;

// JunctionMap::writeToFile (utils/JunctionMap.cpp:579-596), Junction::toString (utils/Junction.cpp:74-89):
//   "<kmer> d0 d1 d2 d3 d4  c0 c1 c2 c3 <sum>  l0 l1 l2 l3 l4 \n"
// in the order given (indices into keys / recs), formatted by hand: a million lines, a few threads, each its own stretch of the order.
int write_junctions(const std::string& path, const uint64_t* keys, const fgpu_junction* recs, const std::vector<uint32_t>& order, int k) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { fprintf(stderr, "cannot write %s\n", path.c_str()); return 2; }
    const size_t kLine = (size_t)k + 1 + 5 * 4 + 1 + 4 * 4 + 5 + 1 + 5 * 2 + 1;   // longest line: three digits and a blank per byte field
    auto format = [&](size_t from, size_t to, std::vector<char>& buf) {
        buf.resize((to - from) * kLine);
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
        buf.resize((size_t)(w - buf.data()));
    };
    const size_t kThreads = 4, kRound = 1u << 20;   // lines per thread and round
    std::vector<char> bufs[kThreads];
    for (size_t base = 0; base < order.size(); base += kThreads * kRound) {
        std::thread workers[kThreads];
        size_t used = 0;
        for (size_t t = 0; t < kThreads; t++) {
            const size_t from = std::min(order.size(), base + t * kRound), to = std::min(order.size(), from + kRound);
            if (from == to) break;
            used = t + 1;
            if (t) workers[t] = std::thread(format, from, to, std::ref(bufs[t]));
            else format(from, to, bufs[0]);
        }
        for (size_t t = 0; t < used; t++) {
            if (workers[t].joinable()) workers[t].join();
            if (fwrite(bufs[t].data(), 1, bufs[t].size(), f) != bufs[t].size()) { fprintf(stderr, "cannot write %s\n", path.c_str()); fclose(f); return 2; }
        }
    }
    fclose(f);
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

int load_pair_filter(PairFilter& pf, const std::string& path) {   // Bloom::load (utils/Bloom.cpp:580-587), with the size checked
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path.c_str()); return 2; }
    const size_t got = fread(pf.bits.data(), 1, pf.bits.size(), f);
    const bool more = fgetc(f) != EOF;
    fclose(f);
    if (got != pf.bits.size() || more) { fprintf(stderr, "%s is not %llu bytes\n", path.c_str(), (unsigned long long)pf.bits.size()); return 2; }
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
        printf("Bits per kmer: %d \n", bits);
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
    // scanInputRead's lists feed the pair filters and the pair counts: the short filter's adds are order-free and happen on the device
    // (fgpu_scan_short_pairs); the lists come to the host only for the paired-end loop (long filter: check-then-insert in file order)
    const bool record_lists = !o.no_cleaning || o.paired_ends;
    const bool device_short_pairs = !o.no_cleaning;
    const bool want_lists = o.paired_ends;
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
    {
        std::thread pin;   // (joined before anything else can fail or read)
        if (!o.batch_reads && !o.from_junctions)
            pin = std::thread([&o] { for (int i = 0; i < 2; i++) pinned_slot(i, kTextPad + o.chunk_bytes); });
        int rc = fgpu_create(&prm, &ctx);
        if (pin.joinable()) pin.join();
        if (rc != FGPU_OK) { fprintf(stderr, "fgpu_create failed (%d): %s\n", rc, fgpu_last_error(nullptr)); return 2; }
    }
    clk.mark("arguments, sizing, fgpu_create");
    std::vector<uint8_t> bloom_bytes(tai / 8);
    std::thread bloom_writer;
    bool bloom_write_failed = false;
    struct JoinWriter {      // every way out of main waits for the .bloom file to be complete
        std::thread& t;
        ~JoinWriter() { if (t.joinable()) t.join(); }
    } join_writer{bloom_writer};

    // ---- pass 1 (load_two_filters, utils/Bloom.cpp:267-350) or -bloom_file (Bloom::load, :580-587)
    if (o.from_bloom) {
        FILE* f = fopen(o.bloom_input_file.c_str(), "rb");
        if (!f) { fprintf(stderr, "cannot open %s\n", o.bloom_input_file.c_str()); return 2; }
        printf("loading bloom filter from file, nelem %llu \n", (unsigned long long)(tai / 8));
        size_t got = fread(bloom_bytes.data(), 1, bloom_bytes.size(), f);
        fclose(f);
        if (got != bloom_bytes.size()) { fprintf(stderr, "%s is not %llu bytes\n", o.bloom_input_file.c_str(), (unsigned long long)(tai / 8)); return 2; }
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
    if (o.just_load) { fgpu_destroy(ctx); return 0; }

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
        fgpu_destroy(ctx);
        fprintf(stderr, "The contig-graph stage is not part of this build: -bloom_file / -junctions_file inputs have been read and checked; the\n"
                        "reference continues from them.\n");
        return 3;
    }

    // ---- pass 2 (ReadScanner::scanReads, src/ReadScanner.cpp:284-359; printScanSummary :19-27)
    {
        time_t start, stop;
        time(&start);
        float w2 = 0;
        CHECK(fgpu_bloom_weight(ctx, FGPU_BLOO2, &w2));
        printf("Weight before read scan: %f \n", w2);
        PairLogic pairs;
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
            int rc = device_short_pairs ? fgpu_scan_short_pairs(ctx, short_pf.tai, short_pf.n_hash, want_lists ? 1 : 0) : FGPU_OK;
            if (rc != FGPU_OK) return failed("fgpu_scan_short_pairs", rc);
            rc = fgpu_scan_begin(ctx);
            if (rc != FGPU_OK) return failed("fgpu_scan_begin", rc);
            uint64_t scanned = 0;
            pairs = PairLogic();
            pairs.k = o.k;
            pairs.paired_ends = o.paired_ends;
            pairs.no_cleaning = o.no_cleaning;
            pairs.short_pf = device_short_pairs ? nullptr : &short_pf;   // (nullptr: the device keeps that filter)
            pairs.long_pf = o.paired_ends ? &long_pf : nullptr;
            std::fill(short_pf.bits.begin(), short_pf.bits.end(), 0);
            std::fill(long_pf.bits.begin(), long_pf.bits.end(), 0);
            std::vector<uint64_t> batch_n_reads;
            // The lists are applied to the long pair filter by a worker thread, in batch order, while this thread goes on feeding the device
            // (the rules are sequential -- check, then insert -- but they need nothing from the scan except the lists).
            struct ListWorker {
                PairLogic& pairs;
                double& busy_ms;
                std::mutex m;
                std::condition_variable cv;
                std::deque<std::pair<std::vector<fgpu_stop>, uint64_t> > q;
                std::vector<std::vector<fgpu_stop> > spare;
                std::vector<fgpu_stop> take_spare() {
                    std::lock_guard<std::mutex> g(m);
                    std::vector<fgpu_stop> v;
                    if (!spare.empty()) { v = std::move(spare.back()); spare.pop_back(); }
                    return v;
                }
                bool closing = false;
                std::thread t;
                ListWorker(PairLogic& p, double& ms) : pairs(p), busy_ms(ms), t([this] { run(); }) {}
                // The worker probes the long pair filter (tens of MB) at random for the whole pass: it stays with the cores that share the
                // last-level cache it starts on, so that the filter stays in that cache instead of following the thread around the machine.
                static void stay_with_this_cache() {
                    const int cpu = sched_getcpu();
                    if (cpu < 0) return;
                    char path[128];
                    snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", cpu);
                    FILE* f = fopen(path, "r");
                    if (!f) return;
                    char list[512] = {0};
                    const size_t got = fread(list, 1, sizeof(list) - 1, f);
                    fclose(f);
                    if (!got) return;
                    cpu_set_t set;
                    CPU_ZERO(&set);
                    int n_set = 0;
                    for (char* p2 = list; *p2;) {           // "0-7,128-135"
                        char* e;
                        const long a = strtol(p2, &e, 10);
                        if (e == p2) break;
                        long b = a;
                        if (*e == '-') { p2 = e + 1; b = strtol(p2, &e, 10); }
                        for (long c = a; c <= b && c < CPU_SETSIZE; c++) { CPU_SET((int)c, &set); n_set++; }
                        p2 = (*e == ',') ? e + 1 : e;
                        if (*e != ',') break;
                    }
                    if (n_set) (void)pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
                }
                void run() {
                    if (!getenv("FGPU_CLI_NO_PIN")) stay_with_this_cache();
                    for (;;) {
                        std::pair<std::vector<fgpu_stop>, uint64_t> item;
                        {
                            std::unique_lock<std::mutex> g(m);
                            cv.wait(g, [&] { return closing || !q.empty(); });
                            if (q.empty()) return;
                            item = std::move(q.front());
                        }
                        const auto t0 = std::chrono::steady_clock::now();
                        pairs.batch(item.first.data(), item.first.size(), item.second);
                        busy_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                        {
                            std::lock_guard<std::mutex> g(m);
                            q.pop_front();                    // only now: `finish` waits for an empty queue
                            spare.push_back(std::move(item.first));   // (its pages are mapped: the next batch's lists go into it without page faults)
                        }
                        cv.notify_all();
                    }
                }
                void put(std::vector<fgpu_stop>&& stops, uint64_t n_reads) {
                    std::unique_lock<std::mutex> g(m);
                    cv.wait(g, [&] { return q.size() < 3; });     // at most three batches of lists in flight
                    q.emplace_back(std::move(stops), n_reads);
                    cv.notify_all();
                }
                void finish() {
                    if (!t.joinable()) return;
                    {
                        std::unique_lock<std::mutex> g(m);
                        cv.wait(g, [&] { return q.empty(); });
                        closing = true;
                    }
                    cv.notify_all();
                    t.join();
                }
                ~ListWorker() { finish(); }
            } worker(pairs, clk.pairs_ms);
            // lists of the oldest batch whose walk is done (the newest one keeps walking while the next batch is prepared)
            auto take = [&](bool& got) -> int {
                got = false;
                uint64_t n = 0;
                int64_t seq = -1;
                const auto t_take = std::chrono::steady_clock::now();
                std::vector<fgpu_stop> stops = worker.take_spare();
                int trc = fgpu_scan_take_stops(ctx, nullptr, 0, &n, &seq);
                if (trc == FGPU_ERR_CAPACITY) {
                    stops.resize(n);
                    trc = fgpu_scan_take_stops(ctx, stops.data(), stops.size(), &n, &seq);
                }
                clk.take_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_take).count();
                if (trc != FGPU_OK) return failed("fgpu_scan_take_stops", trc);
                if (seq < 0) return 0;
                got = true;
                stops.resize(n);
                worker.put(std::move(stops), batch_n_reads[(size_t)seq]);
                return 0;
            };
            fgpu_reads r;
            for (int more; (more = src.next(ctx, &r)) != 0;) {
                if (more < 0) return failed("fgpu_text_split", -more);
                const auto t_scan = std::chrono::steady_clock::now();
                if ((rc = fgpu_scan_batch(ctx, &r)) != FGPU_OK) return failed("fgpu_scan_batch", rc);
                clk.scan_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_scan).count();
                batch_n_reads.push_back(r.n_reads);
                if (want_lists && batch_n_reads.size() > 1) {
                    bool got;
                    if (int trc = take(got)) return trc;
                }
                scanned += r.n_reads;
                fprintf(stdout, "\rreads scanned: %lld", (long long)scanned);
                fflush(stdout);
            }
            if ((rc = fgpu_scan_end(ctx, &ss)) != FGPU_OK) return failed("fgpu_scan_end", rc);
            if (device_short_pairs && (rc = fgpu_scan_short_pairs_download(ctx, short_pf.bits.data(), short_pf.bits.size())) != FGPU_OK)
                return failed("fgpu_scan_short_pairs_download", rc);
            if (want_lists) {
                bool got = true;
                while (got)
                    if (int trc = take(got)) return trc;
            }
            worker.finish();
            return 0;
        };
        // (a preview of the junction walk that the library cannot repair is absorbed inside the library: it keeps the scan's batches in HBM and
        // scans them again by itself -- nothing is read twice here, which is what lets both inputs be pipes)
        const int src_rc = scan_once();
        if (src_rc == -1) { fprintf(stderr, "scan failed: %s\n", fgpu_last_error(ctx)); return 2; }
        if (src_rc) return src_rc;
        time(&stop);
        if (clk.on) fprintf(stderr, "[cli]   %.2f ms in fgpu_scan_batch calls, %.2f ms in fgpu_scan_take_stops, %.2f ms applying the lists to the pair filters (worker thread)\n",
                            clk.scan_ms, clk.take_ms, clk.pairs_ms);
        if (clk.on && o.paired_ends) fprintf(stderr, "[cli]   of the worker's time, %.2f ms preparing canonical forms and hashes and speculating the long-pair checks (helper threads)\n", pairs.prepare_ms);
#ifdef FGPU_CLI_PROFILE
        fprintf(stderr, "[cli-profile] batch(): prepare %.1f ms, read offsets %.1f, pair list %.1f, phase 1 %.1f, loop over reads %.1f ms (long_pairs in it: %.1f Mticks)\n",
                pairs.sec_ms[0], pairs.sec_ms[1], pairs.sec_ms[2], pairs.sec_ms[3], pairs.sec_ms[4], pairs.tk_all / 1e6);
        fprintf(stderr, "[cli-profile] long_pairs %.1f Mticks, of which inserts %.1f; first-probe failures %llu, chains %llu, first-end k-mers found paired %llu\n",
                pairs.tk_long / 1e6, pairs.tk_ins / 1e6, pairs.n_first_fail, pairs.n_chain, pairs.n_hit);
#endif
        clk.mark("pass 2 (read + scan)");
        printf("Empty count: %d, not empty count: %d\n", pairs.empty_count, pairs.not_empty_count);
        printf("Reads processed: %llu\n", (unsigned long long)ss.reads_processed);
        printf("Unambiguous reads: %llu\n", (unsigned long long)ss.unambiguous_reads);
        printf("Time in seconds for read scan: %f \n", difftime(stop, start));
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
        // the reference's dump order = the iteration order of its container after these insertions (junction_order.h)
        std::vector<uint32_t> order;
        if (DumpOrder::agrees_with_the_container(keys.data(), (size_t)std::min<uint64_t>(n, 50000))) {
            order = DumpOrder::of(keys.data(), (size_t)n);
        } else {   // a standard library that links its nodes another way: ask the container itself
            std::unordered_map<uint64_t, uint32_t> container;
            for (uint64_t i = 0; i < n; i++) container.insert(std::pair<uint64_t, uint32_t>(keys[i], (uint32_t)i));
            for (const auto& kv : container) order.push_back(kv.second);
        }
        clk.mark("  dump order");
        printf("Writing to junction file\n");
        if (int rc = write_junctions(o.file_prefix + ".junctions", keys.data(), recs.data(), order, o.k)) return rc;
        printf("Done writing to junction file\n");
        clk.mark("junction download + dump");
        if (!o.no_cleaning) {   // src/Faucet.cpp:297-300
            if (int rc = short_pf.dump(o.file_prefix + ".short_pair_filter")) return rc;
            if (o.paired_ends)
                if (int rc = long_pf.dump(o.file_prefix + ".long_pair_filter")) return rc;
        }
        printf("Weight of short pair filter: %f\n", short_pf.weight());
        if (o.paired_ends) printf("Weight of long pair filter: %f\n", long_pf.weight());
        printf("Number of junctions: %llu\n", (unsigned long long)order.size());
    }
    clk.mark("pair filter weights");
    if (!o.no_cleaning)
        fprintf(stderr, "The contig-graph stage is not part of this build: the load and scan outputs have been written; the reference\n"
                        "continues from the same calls through integration/faucet_binding.cpp (INTEGRATION.md).\n");
    if (bloom_writer.joinable()) bloom_writer.join();      // (_exit below skips destructors)
    if (bloom_write_failed) { fprintf(stderr, "cannot write %s.bloom\n", o.file_prefix.c_str()); return 2; }
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
    clk.mark("fgpu_destroy");
    return code;
}
