// text_source.h — file text for the device-side record splitter (fgpu_text_split), and record-aligned cuts of a file for read shards.
//
// The reference reads its input with `while (getline(header)) { getline(sequence); ...; if (fastq) getline, getline; }`
// (utils/Bloom.cpp:280-282,340; src/ReadScanner.cpp:306-308,349).  TextSource only moves file text -- `chunk` bytes at a time through two
// page-locked buffers, read ahead by a thread of its own -- and the library applies that loop on the device; it carries the unconsumed tail
// (an incomplete record) over to the next call.  Works on non-seekable input.  A source may be limited to a byte range [begin, end) of a
// regular file: the range of one read shard (record_cuts below), read by the host thread that drives the shard's GPU.
#pragma once
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "faucet_gpu.h"

namespace faucet_host {

static const uint64_t kTextPad = 16u << 20;   // room in front of a chunk for the unconsumed tail of the previous one

// Regular input files are copied out of a mapping: memcpy from the page cache runs at 2-4 times the rate of pread by as many threads
// (scripts/micro/read_rate.cpp on the GPU box's host: 4 threads 80 against 34 GB/s), and reading was what pass 1 waited for.
// The mapping belongs to the process, not to the pass: taking a 1 GB mapping down costs 45 ms (a quarter of a million page table
// entries), and both passes usually read the same file -- the second finds the pages mapped already.  Released at exit.
// FGPU_CLI_NO_MMAP=1: positioned reads instead (a file that is truncated while the run reads it raises SIGBUS in a mapping, where a read
// would just end).
inline const char* mapped_file(int fd, const struct stat& st) {
    struct Mapped { dev_t dev; ino_t ino; off_t size; const char* p; };
    static std::vector<Mapped> mapped;
    static std::mutex mapped_m;
    if (getenv("FGPU_CLI_NO_MMAP") || st.st_size <= 0) return nullptr;
    std::lock_guard<std::mutex> g(mapped_m);
    for (const Mapped& mp : mapped)
        if (mp.dev == st.st_dev && mp.ino == st.st_ino && mp.size == st.st_size) return mp.p;
    void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_SHARED, fd, 0);
    if (m == MAP_FAILED) return nullptr;
    (void)madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
    mapped.push_back(Mapped{st.st_dev, st.st_ino, st.st_size, (const char*)m});
    return (const char*)m;
}

// The two text buffers of the single-device run are pinned (the copy to the device runs at link speed) and belong to the process, not to
// one pass: pinning and unpinning 160 MB costs 30 ms each way, which was a fifth of a pass over a 1 GB file.  main() pins them on a helper
// thread while the context is being created; they are returned when the process ends.
inline char* pinned_slot(int i, uint64_t bytes) {
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

class TextSource {
public:
    // bufs: two buffers of kTextPad + chunk bytes each that the caller owns (a shard's own pair), or nullptr for the process's two
    TextSource(const std::string& path, bool fastq, uint64_t chunk, uint64_t begin = 0, uint64_t end = ~0ULL, char* const* bufs = nullptr)
        : fd_(open(path.c_str(), O_RDONLY)), fastq_(fastq), chunk_(chunk), offset_(begin), end_(end) {
        if (fd_ < 0) return;
        struct stat st;
        regular_ = fstat(fd_, &st) == 0 && S_ISREG(st.st_mode);
        if (regular_) {
            map_ = mapped_file(fd_, st);
            if (map_) map_size_ = (uint64_t)st.st_size;
        }
        for (int i = 0; i < 2; i++) slot_[i].base = bufs ? bufs[i] : pinned_slot(i, kPad + chunk_);
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
    TextSource(const TextSource&) = delete;
    TextSource& operator=(const TextSource&) = delete;
    bool is_open() const { return fd_ >= 0 && slot_[0].base && slot_[1].base; }
    bool regular() const { return regular_; }
    uint64_t bytes_handed_out() const { return handed_; }   // file text the batches returned so far cover (the tail carried over not counted)
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
            if (!reserved_) {     // the chunks start small and double: the library sizes its buffers for the largest once (fgpu_text_reserve)
                fgpu_text_reserve(ctx, chunk_ + (1u << 20));
                reserved_ = true;
            }
            const int rc = fgpu_text_split(ctx, text, n, 0, fastq_ ? 1 : 0, sl.eof ? 1 : 0, out, &used);
            split_ms_ += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_split).count();
            if (rc != FGPU_OK) return -rc;
            const uint64_t left = n - used;
            handed_ += used;
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
    bool reserved_ = false;
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
            size_t want = chunk_bytes(n_chunks_++);
            bool range_ends = false;
            if (end_ != ~0ULL) {                     // a shard's range: the last chunk ends with it
                const uint64_t left = end_ > offset_ ? end_ - offset_ : 0;
                if (left <= want) { want = (size_t)left; range_ends = true; }
            }
            const auto t_read = std::chrono::steady_clock::now();
            const size_t got = want ? fill(sl.base + kPad, want) : 0;
            read_ms_ += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_read).count();
            const bool eof = range_ends || got < want;
            {
                std::lock_guard<std::mutex> g(m_);
                sl.got = got;
                sl.eof = eof;
                sl.full = true;
            }
            cv_.notify_all();
            if (eof) return;
        }
    }
    // `want` bytes from the input, fewer only at its end.  A regular file is read by kReaders threads at once (one thread copies out of
    // the page cache at 7-8 GB/s, which was 2.4 times the time the device needs for the same text); anything else (a pipe, a process
    // substitution) is read in order by this thread alone.
    size_t read_fully(int fd, char* dst, size_t want, off_t at, bool positioned) const {
        if (positioned && map_) {                 // (the file as it was when it was mapped: what lies beyond that size is not looked for)
            if ((uint64_t)at >= map_size_) return 0;
            const size_t n = (size_t)std::min<uint64_t>(want, map_size_ - (uint64_t)at);
            memcpy(dst, map_ + at, n);
            return n;
        }
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
    const char* map_ = nullptr;       // a regular file, mapped (read_fully copies out of it)
    uint64_t map_size_ = 0;
    int fd_;
    bool regular_ = false;
    bool fastq_;
    uint64_t chunk_;
    uint64_t offset_ = 0, end_ = ~0ULL;
    uint64_t handed_ = 0;
    Slot slot_[2];
    int cur_ = 0;
    uint64_t tail_ = 0, n_chunks_ = 0;
    bool finished_ = false, stop_ = false;
    std::thread reader_;
    std::mutex m_;
    std::condition_variable cv_;
};

// ---- record-aligned cuts of a regular file into n file-order shards ---------------------------------------------------------------------------
// The reference's loop takes lines as they come -- it never looks at '>' or '@' -- so a record is `lines_per_record` consecutive lines counted
// from the top of the file and nothing else says where one begins: the cut for shard r is the first record boundary at or behind byte
// r * size / n, found by counting the newlines in front of it (every shard's thread counts its own stretch, a prefix sum joins them).
// `group` records stay together (2 for --paired_ends: reads 2p and 2p + 1 of the scan are a pair, src/ReadScanner.cpp:303-350).
// Returns n + 1 offsets, cuts[0] = 0, cuts[n] = size; shards may be empty.  false: not a regular file, or it cannot be read.
inline bool record_cuts(const std::string& path, int lines_per_record, int group, int n, std::vector<uint64_t>* cuts) {
    struct stat st;
    if (stat(path.c_str(), &st) != 0 || !S_ISREG(st.st_mode)) return false;      // (looked at before it is opened: opening a FIFO would wait for its writer)
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { close(fd); return false; }
    const uint64_t size = (uint64_t)st.st_size;
    cuts->assign((size_t)n + 1, size);
    (*cuts)[0] = 0;
    if (n == 1 || size == 0) { close(fd); return true; }
    const char* map = mapped_file(fd, st);
    std::vector<char> whole;
    if (!map) {                                   // FGPU_CLI_NO_MMAP: read it
        whole.resize((size_t)size);
        uint64_t got = 0;
        while (got < size) {
            const ssize_t r = pread(fd, whole.data() + got, (size_t)(size - got), (off_t)got);
            if (r < 0 && errno == EINTR) continue;
            if (r <= 0) break;
            got += (uint64_t)r;
        }
        if (got != size) { close(fd); return false; }
        map = whole.data();
    }
    close(fd);
    const uint64_t period = (uint64_t)lines_per_record * (uint64_t)(group > 0 ? group : 1);
    std::vector<uint64_t> nominal((size_t)n + 1), lines((size_t)n, 0);
    for (int r = 0; r <= n; r++) nominal[(size_t)r] = size / (uint64_t)n * (uint64_t)r + size % (uint64_t)n * (uint64_t)r / (uint64_t)n;
    nominal[(size_t)n] = size;
    auto count = [&](int r) {
        uint64_t c = 0;
        const char* p = map + nominal[(size_t)r];
        const char* e = map + nominal[(size_t)r + 1];
        while (p < e) {
            const char* q = (const char*)memchr(p, '\n', (size_t)(e - p));
            if (!q) break;
            c++;
            p = q + 1;
        }
        lines[(size_t)r] = c;
    };
    std::vector<std::thread> th;
    for (int r = 1; r < n; r++) th.emplace_back(count, r);
    count(0);
    for (std::thread& t : th) t.join();
    uint64_t before = 0;                          // newlines in [0, nominal[r])
    for (int r = 1; r < n; r++) {
        before += lines[(size_t)r - 1];
        uint64_t pos = nominal[(size_t)r], c = before;
        // the first position >= nominal that starts a line whose index is a multiple of the period
        const bool at_line_start = pos == 0 || map[pos - 1] == '\n';
        if (!(at_line_start && c % period == 0)) {
            for (;;) {
                const char* q = pos < size ? (const char*)memchr(map + pos, '\n', (size_t)(size - pos)) : nullptr;
                if (!q) { pos = size; break; }
                pos = (uint64_t)(q - map) + 1;
                c++;
                if (c % period == 0) break;
            }
        }
        (*cuts)[(size_t)r] = pos;
    }
    for (int r = 1; r <= n; r++) (*cuts)[(size_t)r] = std::max((*cuts)[(size_t)r], (*cuts)[(size_t)r - 1]);   // (monotone: a long record may swallow a nominal cut)
    return true;
}

}  // namespace faucet_host
