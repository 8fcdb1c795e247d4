// shard_host.h — the two passes over N GPUs from ONE process: one host thread per device, reads sharded in file order.
//
// The reference is one process on one core: load_two_filters (utils/Bloom.cpp:267-350, called at src/Faucet.cpp:220) and then
// ReadScanner::scanReads (src/ReadScanner.cpp:284-359, called at src/Faucet.cpp:241-245).  BASELINE.json's north_star keeps that command
// line and shards the reads over the GPUs of a node; this header is the host side of it for C++ callers -- the `faucet` command line
// (`-gpus N`, faucet_main.cpp) and the patch a maintainer links into the reference (integration/faucet_binding.cpp) -- over the C ABI only
// (include/faucet_gpu.h: the per-context calls and the fgpu_group_* exchanges; no HIP here).  faucet_amd/sharded.py is the same protocol for
// one PROCESS per GPU over torch.distributed; DESIGN.md section 5 says why each step is exact.
//
//   shard r = the r-th file-order share of the records (text_source.h, record_cuts), read and driven by thread r on device r
//   pass 1   fix-up protocol (a shard that stays in HBM, at most 4 hash functions): every rank loads its shard alone
//            (FGPU_LOAD_SHARD_TIMES) -> exclusive prefix-OR of the shards' bloo1 over ranks -> fgpu_load_fixup against that prefix on
//            ranks > 0 -> OR-allreduce of bloo2;  presence protocol (otherwise, and with --mercy): presence bitmap of the shard ->
//            exclusive prefix-OR = the carried-in bloo1 -> ordered load on it -> OR-allreduce of bloo2
//   pass 2   rank 0 streams its shard and shows the others its junction table after a quarter of it (the preview their pure stage uses);
//            ranks > 0 run the pure stage of their shard meanwhile (fgpu_scan_prepare), then take the table, the counters and the two
//            pair filters from the rank below, walk, and hand all four on.  The last rank holds the run's junction map and pair filters.
#pragma once
#include <string.h>

#include <atomic>
#include <functional>
#include <mutex>
#include <condition_variable>
#include <string>
#include <thread>
#include <vector>

#include "faucet_gpu.h"
#include "text_source.h"

namespace faucet_host {

struct ShardOptions {
    int n_ranks = 1;
    int transport = FGPU_TRANSPORT_COPY;
    std::vector<int> devices;             // device of every rank (size n_ranks)
    fgpu_params prm;                      // .device is filled in per rank
    bool fastq = false, mercy = false, paired_ends = false, no_cleaning = false;
    uint64_t chunk_bytes = 64u << 20;
    uint64_t short_tai = 0, long_tai = 0; // pair filters (0: none), create_bloom_filter_optimal's sizes (src/Faucet.cpp:266-283)
    int short_hashes = 0, long_hashes = 0;
    bool verbose = false;                 // FGPU_CLI_TIMES: per-rank stage times on stderr
};

struct ShardLoadResult {
    fgpu_load_stats stats;                // summed over the shards = the sequential run's
    float w1 = 0, w2 = 0;                 // Bloom::weight of the run's bloo1 / bloo2
    bool fixup = false;
};

struct ShardScanResult {
    fgpu_scan_stats stats;                // the last rank's = the sequential run's
    uint64_t empty_count = 0, not_empty_count = 0;
};

#define RANK_CHECK(call)                                                                                              \
do {                                                                                                              \
    const int rc__ = (call);                                                                                      \
    if (rc__ != FGPU_OK) { err_[(size_t)r] = std::string(#call) + " failed: " + fgpu_last_error(c); return rc__; } \
} while (0)
#define GROUP_CHECK(call)                                                                                             \
do {                                                                                                              \
    const int rc__ = (call);                                                                                      \
    if (rc__ != FGPU_OK) { err_[(size_t)r] = std::string(#call) + " failed: " + fgpu_group_last_error(group_, r); return rc__; } \
} while (0)
#define RANK_TRY(call)                                  \
do {                                                \
    const int rc__ = (call);                        \
    if (rc__ != FGPU_OK) return rc__;               \
} while (0)

class ShardedRun {
public:
    explicit ShardedRun(const ShardOptions& o) : o_(o), ctx_((size_t)o.n_ranks, nullptr), err_((size_t)o.n_ranks), bufs_((size_t)o.n_ranks) { graveyard_.resize((size_t)std::max(o.n_ranks, 1)); }
    ~ShardedRun() { close(); }
    ShardedRun(const ShardedRun&) = delete;
    ShardedRun& operator=(const ShardedRun&) = delete;

    const std::string& error() const { return error_; }
    fgpu_ctx* ctx(int rank) const { return ctx_[(size_t)rank]; }
    fgpu_ctx* first_ctx() const { return ctx_.front(); }
    fgpu_ctx* last_ctx() const { return ctx_.back(); }
    int n_ranks() const { return o_.n_ranks; }
    // the pair filters of the scan (0 bits: none): create_bloom_filter_optimal's sizes, known once the host has made its own (src/Faucet.cpp:266-283)
    void set_pair_filters(uint64_t short_tai, int short_hashes, uint64_t long_tai, int long_hashes) {
        o_.short_tai = short_tai; o_.short_hashes = short_hashes; o_.long_tai = long_tai; o_.long_hashes = long_hashes;
    }

    // contexts (one per rank, made by the rank's thread) and the group that ties them together
    int create() {
        int rc = fgpu_group_create(o_.n_ranks, o_.transport, &group_);
        if (rc != FGPU_OK) {
            error_ = std::string("fgpu_group_create: ") + (group_ ? fgpu_group_last_error(group_, -1) : "bad arguments");
            return rc;
        }
        return run_ranks([this](int r) -> int {
            fgpu_params p = o_.prm;
            p.device = o_.devices[(size_t)r];
            int rc = fgpu_create(&p, &ctx_[(size_t)r]);
            if (rc != FGPU_OK) { err_[(size_t)r] = std::string("fgpu_create: ") + fgpu_last_error(nullptr); return rc; }
            for (int i = 0; i < 2; i++) {
                bufs_[(size_t)r].p[i] = (char*)fgpu_host_alloc(kTextPad + o_.chunk_bytes);
                if (!bufs_[(size_t)r].p[i]) { err_[(size_t)r] = "page-locked text buffers could not be allocated"; return FGPU_ERR_NOMEM; }
            }
            rc = fgpu_group_attach(group_, r, ctx_[(size_t)r]);
            if (rc != FGPU_OK) err_[(size_t)r] = std::string("fgpu_group_attach: ") + fgpu_group_last_error(group_, r);
            return rc;
        });
    }

    void close() {
        if (group_) { fgpu_group_destroy(group_); group_ = nullptr; }
        // exchange buffers of a pass that FAILED: freed only here, behind fgpu_group_destroy, which has synchronised every rank's stream -- a peer's
        // copy may still have been reading them when their rank gave up
        for (size_t r = 0; r < graveyard_.size() && r < ctx_.size(); r++) {
            for (void* p : graveyard_[r]) if (ctx_[r]) fgpu_device_free(ctx_[r], p);
            graveyard_[r].clear();
        }
        for (size_t r = 0; r < ctx_.size(); r++) {
            if (ctx_[r]) { fgpu_destroy(ctx_[r]); ctx_[r] = nullptr; }
            for (int i = 0; i < 2; i++) if (bufs_[r].p[i]) { fgpu_host_free(bufs_[r].p[i]); bufs_[r].p[i] = nullptr; }
        }
    }

    // ---- pass 1 (load_two_filters, utils/Bloom.cpp:267-350): afterwards every rank holds the run's bloo2, the last rank its bloo1
    int load(const std::string& path, ShardLoadResult* out) {
        std::vector<uint64_t> cuts;
        if (!record_cuts(path, o_.fastq ? 4 : 2, o_.paired_ends ? 2 : 1, o_.n_ranks, &cuts)) {
            error_ = "cannot cut " + path + " into read shards: with several GPUs the input must be a regular file";
            return FGPU_ERR_ARG;
        }
        uint64_t largest = 0;
        for (int r = 0; r < o_.n_ranks; r++) largest = std::max(largest, cuts[(size_t)r + 1] - cuts[(size_t)r]);
        // the fix-up protocol keeps the shard's batches in HBM (about a byte per base with the fail planes) within what the library sets aside for
        // resident batches -- asked of every context, not assumed -- and covers four hash functions; --mercy leaves no fix-up state
        uint64_t budget = ~0ULL;
        for (int r = 0; r < o_.n_ranks; r++) {
            uint64_t b = 0;
            if (fgpu_load_fixup_state(ctx_[(size_t)r], nullptr, &b) != FGPU_OK) b = 0;
            budget = std::min(budget, b);
        }
        const char* force_planes = getenv("FAUCET_SHARD_PLANES");
        const bool short_shards = largest < 0xFFF00000ULL - (1ULL << 24) && !(force_planes && force_planes[0] == '1' && o_.prm.n_hash <= 4);   // a shard's positions (<= its bytes) fit one 32-bit clock
        bool fixup = !o_.mercy && (short_shards || o_.prm.n_hash <= 4) && largest + (64ULL << 20) < budget;
        if (const char* e = getenv("FAUCET_SHARD_PROTOCOL")) {
            if (!strcmp(e, "presence")) fixup = false;
            else if (!strcmp(e, "fixup") && !fixup) { error_ = "FAUCET_SHARD_PROTOCOL=fixup: a shard is too large to stay in HBM, more than 4 hash functions, or --mercy is on"; return FGPU_ERR_ARG; }
        }
        const char* dbg_not_ready = getenv("FAUCET_DEBUG_FIXUP_NOT_READY");      // tests: this rank's own load "did not stay resident"
        std::vector<fgpu_load_stats> st((size_t)o_.n_ranks);
        float w1 = 0, w2 = 0;
        const uint64_t nbytes = o_.prm.tai / 8;
        std::atomic<bool> fixup_done{fixup};
        int rc = run_ranks([&](int r) -> int {
            fgpu_ctx* c = ctx_[(size_t)r];
            const double t0 = now_ms();
            void *b1 = nullptr, *b2 = nullptr, *prefix = nullptr;
            RANK_CHECK(fgpu_device_alloc(c, nbytes, &prefix));
            Deferred free_prefix(this, r, &prefix, nullptr, nullptr, nullptr);
            bool use_fixup = fixup;
            double t1 = t0;
            if (use_fixup) {
                RANK_CHECK(fgpu_load_begin(c, short_shards ? FGPU_LOAD_SHARD_TIMES : FGPU_LOAD_SHARD_PLANES));
                RANK_TRY(for_each_batch(r, path, cuts, [&](const fgpu_reads* b) { return fgpu_load_batch(c, b); }, nullptr));
                RANK_CHECK(fgpu_load_end(c, &st[(size_t)r]));
                t1 = now_ms();
                // Can every rank complete its pass by the fix-up?  (A batch that did not stay resident -- over the budget, or no memory at that
                // moment -- leaves fgpu_load_fixup nothing to work on.)  The ranks agree: if one cannot, ALL run the presence protocol instead --
                // a pass more, the same filters -- rather than the run dying after a full pass 1.
                int ready = 0;
                RANK_CHECK(fgpu_load_fixup_state(c, &ready, nullptr));
                if (dbg_not_ready && atoi(dbg_not_ready) == r) ready = 0;
                if (!vote(ready != 0)) {
                    if (aborted_) { err_[(size_t)r] = "the run was aborted (another rank failed)"; return FGPU_ERR_STATE; }
                    use_fixup = false;
                    fixup_done = false;
                    tell(r, "pass 1: own load %.1f ms; a rank cannot complete it by the fix-up: the presence protocol instead", t1 - t0);
                }
            }
            if (use_fixup) {
                RANK_CHECK(fgpu_bloom_devptr(c, FGPU_BLOO1, &b1, nullptr));
                GROUP_CHECK(fgpu_group_exclusive_prefix_or(group_, r, b1, prefix, nbytes));
                if (r > 0) RANK_CHECK(fgpu_load_fixup(c, prefix, &st[(size_t)r]));
                tell(r, "pass 1: own load %.1f ms, prefix-OR exchange + fix-up %.1f ms", t1 - t0, now_ms() - t1);
            } else {
                const double t1p = now_ms();
                RANK_CHECK(fgpu_load_begin(c, 0));             // (empties both filters)
                RANK_CHECK(fgpu_load_end(c, nullptr));
                RANK_TRY(for_each_batch(r, path, cuts, [&](const fgpu_reads* b) { return fgpu_presence_batch(c, b); }, nullptr));
                const double t2p = now_ms();
                RANK_CHECK(fgpu_bloom_devptr(c, FGPU_BLOO1, &b1, nullptr));
                GROUP_CHECK(fgpu_group_exclusive_prefix_or(group_, r, b1, prefix, nbytes));
                RANK_CHECK(fgpu_device_copy(c, b1, prefix, nbytes));      // the carried-in bloo1 of this shard
                RANK_CHECK(fgpu_load_begin(c, FGPU_LOAD_KEEP_CARRY));
                RANK_TRY(for_each_batch(r, path, cuts, [&](const fgpu_reads* b) { return fgpu_load_batch(c, b); }, nullptr));
                RANK_CHECK(fgpu_load_end(c, &st[(size_t)r]));
                tell(r, "pass 1: presence pass %.1f ms, prefix-OR exchange + ordered load %.1f ms", t2p - t1p, now_ms() - t2p);
            }
            const double t2 = now_ms();
            RANK_CHECK(fgpu_bloom_devptr(c, FGPU_BLOO2, &b2, nullptr));
            GROUP_CHECK(fgpu_group_or_allreduce(group_, r, b2, nbytes));
            if (r == o_.n_ranks - 1) {
                RANK_CHECK(fgpu_bloom_weight(c, FGPU_BLOO1, &w1));
                RANK_CHECK(fgpu_bloom_weight(c, FGPU_BLOO2, &w2));
            } else {
                RANK_CHECK(fgpu_synchronize(c));
            }
            tell(r, "pass 1: OR-allreduce of bloo2 %.1f ms (to the device's completion)", now_ms() - t2);
            free_prefix.ok = true;
            return FGPU_OK;
        });
        if (rc != FGPU_OK) return rc;
        memset(&out->stats, 0, sizeof(out->stats));
        for (const fgpu_load_stats& s : st) {
            out->stats.reads_processed += s.reads_processed;
            out->stats.unambiguous_reads += s.unambiguous_reads;
            out->stats.kmers += s.kmers;
            out->stats.to_bloo2 += s.to_bloo2;
        }
        out->w1 = w1;
        out->w2 = w2;
        out->fixup = fixup_done;
        return FGPU_OK;
    }

    // ---- pass 2 (ReadScanner::scanReads, src/ReadScanner.cpp:284-359): afterwards the last rank holds the junction map and the pair filters
    int scan(const std::string& path, ShardScanResult* out) {
        std::vector<uint64_t> cuts;
        if (!record_cuts(path, o_.fastq ? 4 : 2, o_.paired_ends ? 2 : 1, o_.n_ranks, &cuts)) {
            error_ = "cannot cut " + path + " into read shards: with several GPUs the input must be a regular file";
            return FGPU_ERR_ARG;
        }
        const int n = o_.n_ranks;
        const bool short_pairs = !o_.no_cleaning && o_.short_tai, long_filter = o_.paired_ends && !o_.no_cleaning && o_.long_tai;
        hint_ = Announce();
        chain_.assign((size_t)n, Announce());
        late_.assign((size_t)n, Announce());
        const char* late_env = getenv("FAUCET_LATE_HINT");
        const bool late_hints = !(late_env && late_env[0] == '0');
        std::vector<uint64_t> empty((size_t)n, 0), not_empty((size_t)n, 0);
        fgpu_scan_stats last_stats;
        memset(&last_stats, 0, sizeof(last_stats));
        int rc = run_ranks([&](int r) -> int {
            fgpu_ctx* c = ctx_[(size_t)r];
            const double t0 = now_ms();
            if (short_pairs) RANK_CHECK(fgpu_scan_short_pairs(c, o_.short_tai, o_.short_hashes, 0));
            if (o_.paired_ends)
                RANK_CHECK(long_filter ? fgpu_scan_long_pairs(c, o_.long_tai, o_.long_hashes, FGPU_LONG_PAIRS_FILTER) : fgpu_scan_long_pairs(c, 0, 0, FGPU_LONG_PAIRS_COUNT));
            RANK_CHECK(fgpu_scan_begin(c));
            fgpu_scan_stats st;
            memset(&st, 0, sizeof(st));
            void *hint_buf = nullptr, *table_in = nullptr, *table_out = nullptr, *late_buf = nullptr;
            Deferred free_all(this, r, &hint_buf, &table_in, &table_out, &late_buf);
            if (r == 0) {
                // the first shard has nothing to wait for: it streams (pure stage of batch b + 1 beside the walk of batch b, lazy junction tests).
                // Once a quarter of it is walked the others are shown its table -- an earlier state of the very table they will be handed, which is
                // all the preview of their pure stage needs (fgpu_scan_import_hint); the hint never enters a result.
                bool shown = n == 1;
                const uint64_t quarter = (cuts[1] - cuts[0]) / 4;
                auto show = [&]() -> int {
                    uint64_t n_entries = 0;
                    RANK_CHECK(fgpu_scan_table_entries(c, &n_entries));
                    RANK_CHECK(fgpu_device_alloc(c, std::max<uint64_t>(n_entries, 1) * FGPU_TABLE_ENTRY_BYTES, &hint_buf));
                    uint64_t got = 0;
                    RANK_CHECK(fgpu_scan_export_table(c, hint_buf, std::max<uint64_t>(n_entries, 1) * FGPU_TABLE_ENTRY_BYTES, &got));
                    for (int q = 1; q < n; q++) GROUP_CHECK(fgpu_group_send_async(group_, 0, q, hint_buf, got * FGPU_TABLE_ENTRY_BYTES));
                    announce(&hint_, got, nullptr);
                    shown = true;
                    return FGPU_OK;
                };
                // Round 6: the rank above this one has nobody to pass it a table, so it is shown a second, later state of this one (after 7/10 of the
                // shard): its planes and candidate planes are made against a table a few million keys short of the one it is handed, and its
                // hop is the short one of every later rank (fgpu_scan_refresh_prepared).  Sent without waiting: the scan goes on beside the copy.
                bool shown_late = n == 1 || !late_hints;
                const uint64_t late_at = (cuts[1] - cuts[0]) / 10 * 7;
                auto show_late = [&]() -> int {
                    uint64_t n_entries = 0;
                    RANK_CHECK(fgpu_scan_table_entries(c, &n_entries));
                    RANK_CHECK(fgpu_device_alloc(c, std::max<uint64_t>(n_entries, 1) * FGPU_TABLE_ENTRY_BYTES, &late_buf));
                    uint64_t got = 0;
                    RANK_CHECK(fgpu_scan_export_table(c, late_buf, std::max<uint64_t>(n_entries, 1) * FGPU_TABLE_ENTRY_BYTES, &got));
                    GROUP_CHECK(fgpu_group_send_async(group_, 0, 1, late_buf, got * FGPU_TABLE_ENTRY_BYTES));
                    announce(&late_[1], got, nullptr);
                    shown_late = true;
                    return FGPU_OK;
                };
                RANK_TRY(for_each_batch(r, path, cuts, [&](const fgpu_reads* b) { return fgpu_scan_batch(c, b); },
                                        [&](uint64_t bytes_done) -> int {
                                            if (!shown && bytes_done >= quarter) { const int src = show(); if (src != FGPU_OK) return src; }
                                            if (shown && !shown_late && bytes_done >= late_at) return show_late();
                                            return FGPU_OK;
                                        }));
                if (!shown) RANK_TRY(show());                 // (a shard without batches still owes the others their preview)
                if (!shown_late) RANK_TRY(show_late());
                RANK_CHECK(fgpu_scan_end(c, &st));
                tell(r, "pass 2: first shard streamed in %.1f ms", now_ms() - t0);
            } else {
                // pure stage while the lower shards walk.  It does not wait for the preview: batches prepared before it arrives see an empty table
                // (every junction test evaluated), the others the preview.
                bool have_hint = false;
                auto take_hint = [&](bool wait, bool use) -> int {
                    uint64_t n_entries = 0;
                    if (!announced(&hint_, wait, &n_entries, nullptr)) return aborted_ ? FGPU_ERR_STATE : FGPU_OK;
                    RANK_CHECK(fgpu_device_alloc(c, std::max<uint64_t>(n_entries, 1) * FGPU_TABLE_ENTRY_BYTES, &hint_buf));
                    GROUP_CHECK(fgpu_group_recv(group_, r, 0, hint_buf, n_entries * FGPU_TABLE_ENTRY_BYTES));
                    if (use) RANK_CHECK(fgpu_scan_import_hint(c, hint_buf, n_entries));
                    have_hint = true;
                    return FGPU_OK;
                };
                RANK_TRY(for_each_batch(r, path, cuts, [&](const fgpu_reads* b) -> int {
                    if (!have_hint) { const int rc = take_hint(false, true); if (rc != FGPU_OK) return rc; }
                    return fgpu_scan_prepare(c, b);
                }, nullptr));
                if (!have_hint) RANK_TRY(take_hint(true, false));   // the send is received even when it came too late to be of use
                if (late_hints && r >= 1) {
                    // the table the rank below has just been HANDED, passed on at once: a preview one shard older than the table this rank will get
                    // (the rank above the first one: the first rank's own table after 7/10 of its shard).
                    // The planes of the prepared batches are made again against it while the rank below walks; the walk then only looks for the
                    // keys that rank created (fgpu_scan_refresh_prepared, faucet_gpu.h)
                    uint64_t n_late = 0;
                    if (!announced(&late_[(size_t)r], true, &n_late, nullptr)) { err_[(size_t)r] = "the run was aborted while this rank waited for the fresher preview"; return FGPU_ERR_STATE; }
                    RANK_CHECK(fgpu_device_alloc(c, std::max<uint64_t>(n_late, 1) * FGPU_TABLE_ENTRY_BYTES, &late_buf));
                    GROUP_CHECK(fgpu_group_recv(group_, r, r - 1, late_buf, n_late * FGPU_TABLE_ENTRY_BYTES));
                    RANK_CHECK(fgpu_scan_import_hint(c, late_buf, n_late));
                    RANK_CHECK(fgpu_scan_refresh_prepared(c));
                }
                const double t1 = now_ms();
                uint64_t n_in = 0;
                fgpu_scan_stats carried;
                if (!announced(&chain_[(size_t)r], true, &n_in, &carried)) { err_[(size_t)r] = "the run was aborted while this rank waited for the junction table"; return FGPU_ERR_STATE; }
                RANK_CHECK(fgpu_device_alloc(c, std::max<uint64_t>(n_in, 1) * FGPU_TABLE_ENTRY_BYTES, &table_in));
                GROUP_CHECK(fgpu_group_recv(group_, r, r - 1, table_in, n_in * FGPU_TABLE_ENTRY_BYTES));
                if (late_hints && r + 1 < n) {                       // pass it on before walking on it: the copy runs beside the walk
                    GROUP_CHECK(fgpu_group_send_async(group_, r, r + 1, table_in, n_in * FGPU_TABLE_ENTRY_BYTES));
                    announce(&late_[(size_t)r + 1], n_in, nullptr);
                }
                RANK_TRY(move_pair_filters(r, short_pairs, long_filter, false));
                const double t2 = now_ms();
                RANK_CHECK(fgpu_scan_import_table(c, table_in, n_in, &carried));   // (takes the place of the preview)
                RANK_CHECK(fgpu_scan_walk_prepared(c));
                RANK_CHECK(fgpu_scan_end(c, &st));
                tell(r, "pass 2: pure stage %.1f ms, waited %.1f ms for the table, import + walk %.1f ms", t1 - t0, t2 - t1, now_ms() - t2);
            }
            if (o_.paired_ends) RANK_CHECK(fgpu_scan_long_pairs_download(c, nullptr, 0, &empty[(size_t)r], &not_empty[(size_t)r]));
            if (r < n - 1) {
                uint64_t n_out = 0, got = 0;
                RANK_CHECK(fgpu_scan_table_entries(c, &n_out));
                RANK_CHECK(fgpu_device_alloc(c, std::max<uint64_t>(n_out, 1) * FGPU_TABLE_ENTRY_BYTES, &table_out));
                RANK_CHECK(fgpu_scan_export_table(c, table_out, std::max<uint64_t>(n_out, 1) * FGPU_TABLE_ENTRY_BYTES, &got));
                announce(&chain_[(size_t)r + 1], got, &st);
                GROUP_CHECK(fgpu_group_send(group_, r, r + 1, table_out, got * FGPU_TABLE_ENTRY_BYTES));
                RANK_TRY(move_pair_filters(r, short_pairs, long_filter, true));
            } else {
                last_stats = st;
            }
            if (n > 1 && (r == 0 || (late_hints && r + 1 < n))) GROUP_CHECK(fgpu_group_flush(group_, r));      // (asynchronous sends: the previews)
            RANK_CHECK(fgpu_synchronize(c));
            free_all.ok = true;
            return FGPU_OK;
        });
        if (rc != FGPU_OK) return rc;
        out->stats = last_stats;
        out->empty_count = out->not_empty_count = 0;
        for (int r = 0; r < n; r++) { out->empty_count += empty[(size_t)r]; out->not_empty_count += not_empty[(size_t)r]; }
        return FGPU_OK;
    }

private:
    struct Bufs { char* p[2] = {nullptr, nullptr}; };
    // device buffers of one rank's pass: freed at once when the pass ended well (everything that touched them has been waited for), else left to close()
    struct Deferred {
        ShardedRun* run;
        int r;
        void** p[4];
        bool ok;
        Deferred(ShardedRun* run_, int r_, void** a, void** b, void** c, void** d) : run(run_), r(r_), ok(false) { p[0] = a; p[1] = b; p[2] = c; p[3] = d; }
        Deferred(const Deferred&) = delete;
        ~Deferred() {
            for (void** q : p) {
                if (!q || !*q) continue;
                if (ok) fgpu_device_free(run->ctx_[(size_t)r], *q);
                else { std::lock_guard<std::mutex> g(run->m_); run->graveyard_[(size_t)r].push_back(*q); }
            }
        }
    };
    // every rank arrives with its answer; true iff all said yes (false as well when the run has been aborted meanwhile)
    bool vote(bool mine) {
        std::unique_lock<std::mutex> g(m_);
        const uint64_t gen = vote_gen_;
        vote_all_ = vote_all_ && mine;
        if (++vote_count_ == o_.n_ranks) {
            vote_result_ = vote_all_;
            vote_count_ = 0;
            vote_all_ = true;
            vote_gen_++;
            cv_.notify_all();
            return vote_result_;
        }
        cv_.wait(g, [&] { return vote_gen_ != gen || aborted_; });
        return vote_gen_ != gen ? vote_result_ : false;
    }
    // what one rank tells another through host memory (the ranks share a process): a count, the scan's counters.  The bytes themselves
    // travel device to device (fgpu_group_send / _recv).
    struct Announce {
        bool ready = false;
        uint64_t n = 0;
        fgpu_scan_stats stats;
        Announce() { memset(&stats, 0, sizeof(stats)); }
    };

    static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

    template <class... A>
    void tell(int r, const char* fmt, A... a) const {
        if (!o_.verbose) return;
        char b[320];
        snprintf(b, sizeof(b), fmt, a...);
        fprintf(stderr, "[cli] rank %d %s\n", r, b);
    }

    void announce(Announce* a, uint64_t n, const fgpu_scan_stats* st) {
        {
            std::lock_guard<std::mutex> g(m_);
            a->n = n;
            if (st) a->stats = *st;
            a->ready = true;
        }
        cv_.notify_all();
    }
    bool announced(Announce* a, bool wait, uint64_t* n, fgpu_scan_stats* st) {
        std::unique_lock<std::mutex> g(m_);
        if (wait) cv_.wait(g, [&] { return a->ready || aborted_; });
        if (!a->ready) return false;
        *n = a->n;
        if (st) *st = a->stats;
        return true;
    }

    // the two pair filters travel with the junction table: what rank r ends with is what rank r + 1 starts from (adds only; check-then-insert
    // in file order; the shards begin at even records, so no first end is left waiting across a cut)
    int move_pair_filters(int r, bool short_pairs, bool long_filter, bool send) {
        fgpu_ctx* c = ctx_[(size_t)r];
        for (int which = 0; which < 2; which++) {
            if (which == 0 ? !short_pairs : !long_filter) continue;
            void* p = nullptr;
            uint64_t nb = 0;
            RANK_CHECK(fgpu_scan_pairs_devptr(c, which, &p, &nb));
            if (send) GROUP_CHECK(fgpu_group_send(group_, r, r + 1, p, nb));
            else GROUP_CHECK(fgpu_group_recv(group_, r, r - 1, p, nb));
        }
        return FGPU_OK;
    }

    // the batches of rank r's shard, in file order; after(bytes of the shard handed out so far) runs behind each
    int for_each_batch(int r, const std::string& path, const std::vector<uint64_t>& cuts, const std::function<int(const fgpu_reads*)>& each,
                       const std::function<int(uint64_t)>& after) {
        fgpu_ctx* c = ctx_[(size_t)r];
        TextSource src(path, o_.fastq, o_.chunk_bytes, cuts[(size_t)r], cuts[(size_t)r + 1], bufs_[(size_t)r].p);
        if (!src.is_open()) { err_[(size_t)r] = "cannot open " + path; return FGPU_ERR_ARG; }
        fgpu_reads b;
        for (int more; (more = src.next(c, &b)) != 0;) {
            if (more < 0) { err_[(size_t)r] = std::string("fgpu_text_split failed: ") + fgpu_last_error(c); return -more; }
            const int rc = each(&b);
            if (rc != FGPU_OK) { if (err_[(size_t)r].empty()) err_[(size_t)r] = std::string("a batch call failed: ") + fgpu_last_error(c); return rc; }
            if (after) { const int rc2 = after(src.bytes_handed_out()); if (rc2 != FGPU_OK) return rc2; }
            if (aborted_) { err_[(size_t)r] = "the run was aborted (another rank failed)"; return FGPU_ERR_STATE; }
        }
        return FGPU_OK;
    }

    // one thread per rank; the first failure aborts the group (every thread that waits for another wakes up and fails) and is reported
    int run_ranks(const std::function<int(int)>& body) {
        std::vector<int> rcs((size_t)o_.n_ranks, FGPU_OK);
        std::vector<std::thread> th;
        for (int r = 0; r < o_.n_ranks; r++)
            th.emplace_back([&, r] {
                rcs[(size_t)r] = body(r);
                if (rcs[(size_t)r] != FGPU_OK) {
                    {
                        std::lock_guard<std::mutex> g(m_);
                        aborted_ = true;
                    }
                    cv_.notify_all();
                    if (group_) fgpu_group_abort(group_);
                }
            });
        for (std::thread& t : th) t.join();
        // the rank that failed first by its own doing (not by the abort) is the one to report
        int rc = FGPU_OK;
        for (int pass = 0; pass < 2 && rc == FGPU_OK; pass++)
            for (int r = 0; r < o_.n_ranks; r++) {
                if (rcs[(size_t)r] == FGPU_OK) continue;
                const bool by_abort = err_[(size_t)r].find("abort") != std::string::npos;
                if (pass == 0 && by_abort) continue;
                rc = rcs[(size_t)r];
                error_ = "rank " + std::to_string(r) + ": " + err_[(size_t)r];
                break;
            }
        return rc;
    }

#undef RANK_CHECK
#undef GROUP_CHECK
#undef RANK_TRY

    ShardOptions o_;
    fgpu_group* group_ = nullptr;
    std::vector<fgpu_ctx*> ctx_;
    std::vector<std::string> err_;
    std::vector<Bufs> bufs_;
    std::string error_;
    std::mutex m_;
    std::condition_variable cv_;
    std::atomic<bool> aborted_{false};
    Announce hint_;
    std::vector<Announce> chain_;
    std::vector<Announce> late_;      // the table a rank was handed, passed on to the rank above as a fresher preview
    std::vector<std::vector<void*>> graveyard_{64};
    int vote_count_ = 0;
    bool vote_all_ = true, vote_result_ = false;
    uint64_t vote_gen_ = 0;
};

}  // namespace faucet_host
