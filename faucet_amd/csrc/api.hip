// api.hip — the extern "C" entry points of include/faucet_gpu.h.
//
// Thin: argument checks, state machine (idle -> loading -> idle -> scanning -> idle), buffer management and the
// order in which the stages of pack.hip / load.hip / scan_pure.hip / scan_walk.hip are put on the stream.
// There is no CPU path in this library: with no usable gfx950 device fgpu_create fails with FGPU_ERR_HIP.
#include <cstdio>
#include <algorithm>
#include <string>

#include <chrono>
#include <thread>

#include "fgpu_ctx.h"

int fgpu_scan_export_impl(fgpu_ctx* ctx, void* dev_entries, uint64_t cap_entries, uint64_t* d_stamps, uint64_t* n_entries);
int fgpu_scan_import_impl(fgpu_ctx* ctx, const void* dev_entries, uint64_t n);
int fgpu_scan_download_impl(fgpu_ctx* ctx, uint64_t* keys_host, fgpu_junction* recs_host, uint64_t cap, uint64_t* n_out);

static thread_local std::string g_create_error;   // (per thread: the ranks of a group create their contexts at the same time)

// ---- helpers declared in fgpu_ctx.h ---------------------------------------------------------------------------
int g_fgpu_trace = getenv("FGPU_TRACE") && getenv("FGPU_TRACE")[0] == '1';

int fgpu_ensure(fgpu_ctx* ctx, DevBuf* b, uint64_t bytes) {
    if (b->bytes >= bytes && b->p) return FGPU_OK;
    if (b->p) {
        FGPU_HIP(fgpu_sync_stream(ctx, ctx->stream));
        FGPU_HIP(hipFree(b->p));
        b->p = nullptr;
        b->bytes = 0;
    } else {
        ctx->owned.push_back(b);
    }
    uint64_t want = bytes + bytes / 8 + 256;   // head room so that slightly larger batches do not reallocate
    hipError_t e = hipMalloc(&b->p, want);
    if (e != hipSuccess) {
        ctx->err = std::string("hipMalloc of ") + std::to_string(want) + " bytes failed: " + hipGetErrorString(e);
        b->p = nullptr;
        return FGPU_ERR_NOMEM;
    }
    b->bytes = want;
    return FGPU_OK;
}

int fgpu_ensure_b(fgpu_ctx* ctx, DevBuf* b, uint64_t bytes) {
    if (b->bytes >= bytes && b->p) return FGPU_OK;
    const double s = ctx->ensure_scale > 1.0 ? ctx->ensure_scale : 1.0;
    const uint64_t want = (uint64_t)((double)bytes * s);
    if (fgpu_ensure(ctx, b, want) == FGPU_OK) return FGPU_OK;
    (void)hipGetLastError();                 // no room for the largest batch: what this one needs
    return fgpu_ensure(ctx, b, bytes);
}

int fgpu_prof_begin(fgpu_ctx* ctx, const char* name) {
    if (!ctx->profile || ctx->prof_suppress) return -1;
    int idx = -1;
    for (size_t i = 0; i < ctx->kstats.size(); i++)
        if (ctx->kstats[i].name == name) { idx = (int)i; break; }
    if (idx < 0) {
        KernelStat ks;
        ks.name = name;
        ctx->kstats.push_back(ks);
        idx = (int)ctx->kstats.size() - 1;
    }
    PendingEvent pe;
    pe.stat = idx;
    if (hipEventCreate(&pe.a) != hipSuccess || hipEventCreate(&pe.b) != hipSuccess) return -1;
    hipEventRecord(pe.a, ctx->launch_stream);
    ctx->pending_events.push_back(pe);
    return (int)ctx->pending_events.size() - 1;
}

void fgpu_prof_end(fgpu_ctx* ctx, int token) {
    if (token < 0) return;
    hipEventRecord(ctx->pending_events[token].b, ctx->launch_stream);
}

// the background part of fgpu_create has finished (and reports here if it failed)
int fgpu_bg_join(fgpu_ctx* ctx) {
    if (ctx->bg) {
        ctx->bg->join();
        delete ctx->bg;
        ctx->bg = nullptr;
    }
    if (ctx->bg_rc) { ctx->err = ctx->bg_err; return ctx->bg_rc; }
    return FGPU_OK;
}

int fgpu_prof_collect(fgpu_ctx* ctx) {
    if (ctx->pending_events.empty()) return FGPU_OK;
    if (int rc = fgpu_bg_join(ctx)) return rc;
    FGPU_HIP(fgpu_sync_stream(ctx, ctx->wstream));
    FGPU_HIP(fgpu_sync_stream(ctx, ctx->stream));
    for (PendingEvent& pe : ctx->pending_events) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, pe.a, pe.b) == hipSuccess) {
            ctx->kstats[pe.stat].launches++;
            ctx->kstats[pe.stat].total_ms += ms;
        }
        hipEventDestroy(pe.a);
        hipEventDestroy(pe.b);
    }
    ctx->pending_events.clear();
    return FGPU_OK;
}

static int check_errors(fgpu_ctx* ctx) {
    // device-side error flags (table overflow) are surfaced at the synchronising calls
    if (ctx->counters_host->error_flags & 1ULL) {
        // A batch created more records than the table had room for (it is kept below a quarter full BETWEEN batches).  While the scan's batches
        // are all in the journal the library absorbs that itself (round 6): the next entry point scans the journal again
        // on a table four times the size (scan_replay) -- the reference's unordered_map simply grows (utils/JunctionMap.h:61).
        if (ctx->journal_on && ctx->phase == 2 && !ctx->in_replay && ctx->jcap < (1ULL << 31)) {
            ctx->lazy_failed = ctx->capacity_failed = true;
            return FGPU_OK;
        }
        ctx->err = "junction table full: raise fgpu_params.junction_capacity";
        return FGPU_ERR_CAPACITY;
    }
    if (ctx->counters_host->error_flags & 2ULL) { ctx->err = "window table full"; return FGPU_ERR_CAPACITY; }
    if (ctx->counters_host->error_flags & 32ULL) { ctx->err = "fgpu_reads.total_bases does not match the batch's offsets"; return FGPU_ERR_ARG; }
    // FGPU_DEBUG_LAZY_FAIL=1 pretends the self-check of the lazy flags fired (tests of the callers' fall-back to eager flags)
    // (= n > 1: only once n batches of the scan have been prepared, so that the replay finds batches whose lists have been harvested already)
    static const long lazy_fail_knob = getenv("FGPU_DEBUG_LAZY_FAIL") ? atol(getenv("FGPU_DEBUG_LAZY_FAIL")) : 0;
    const bool force_lazy_fail = lazy_fail_knob == 1 || (lazy_fail_knob > 1 && ctx->scan_batch_index >= (uint64_t)lazy_fail_knob);
    const bool lazy = !(ctx->prm.flags & FGPU_FLAG_EAGER_FLAGS) && !ctx->eager_runtime && !ctx->eager_scan;
    // (before the key-ordered walk's own flags: a late junction test that comes out true inside a large cluster voids the lazy scan
    // -- bit 4 -- and the piece then stops at a k-mer nobody registered -- bit 16: a consequence, gone with the eager scan that follows)
    if ((ctx->counters_host->error_flags & 4ULL) || (force_lazy_fail && lazy && ctx->phase == 2)) {
        if (ctx->journal_on && ctx->phase == 2 && !ctx->in_replay) {
            ctx->lazy_failed = true;     // every batch of this scan is still in HBM: the next entry point scans them again, eagerly (scan_replay)
            return FGPU_OK;
        }
        ctx->err = "lazy-flag check failed: the walk scanned a position whose junction test was not evaluated; "
                   "repeat the scan after fgpu_scan_set_eager(ctx, 1) (or with FGPU_FLAG_EAGER_FLAGS)";
        return FGPU_ERR_STATE;
    }
    if (ctx->counters_host->error_flags & 8ULL) { ctx->err = "key-ordered walk: a k-mer's turn never came (internal error)"; return FGPU_ERR_STATE; }
    if (ctx->counters_host->error_flags & 16ULL) { ctx->err = "key-ordered walk: a junction at a k-mer nobody registered (internal error)"; return FGPU_ERR_STATE; }
    return FGPU_OK;
}

// everything issued so far, on both streams, has completed
static int sync_all(fgpu_ctx* ctx) {
    if (int rc = fgpu_bg_join(ctx)) return rc;
    FGPU_HIP(fgpu_sync_stream(ctx, ctx->wstream));
    FGPU_HIP(fgpu_sync_stream(ctx, ctx->cstream));   // the side stream's last resets (nothing waits for them but the next window of their parity)
    FGPU_HIP(fgpu_sync_stream(ctx, ctx->stream));
    return FGPU_OK;
}

static int pull_counters(fgpu_ctx* ctx) {
    if (int rc = fgpu_bg_join(ctx)) return rc;
    FGPU_HIP(fgpu_sync_stream(ctx, ctx->wstream));
    FGPU_HIP(fgpu_sync_stream(ctx, ctx->cstream));
    FGPU_HIP(hipMemcpyAsync(ctx->counters_host, ctx->counters, sizeof(DevCounters), hipMemcpyDeviceToHost, ctx->stream));
    FGPU_HIP(fgpu_sync_stream(ctx, ctx->stream));
    // A late junction test that came out true at an unregistered k-mer leaves the lazy scan standing only because k_delta_collect has looked
    // for that k-mer on the window's other pieces.  Every walk and every sweep issued so far has completed here, so each
    // noted position must have been passed over by its window's sweep -- [2] == [0]; if one was not, the scan is treated as void and scanned
    // again eagerly rather than trusted (the self-test as a run-time guard, not only a diagnostic).
    if (ctx->phase == 2) {
        DevCounters& c = *ctx->counters_host;
        if (c.late_n[0] <= FGPU_LATE_CAP && c.late_n[2] != c.late_n[0]) c.error_flags |= 4ULL;
    }
    return check_errors(ctx);
}

// the page-locked buffers of harvested lists go back to the pool (stop_queue.clear() would leak them)
static void stop_queue_recycle(fgpu_ctx* ctx) {
    for (StopBatch& sb : ctx->stop_queue)
        if (sb.data) { StopBatch f; f.data = sb.data; f.cap = sb.cap; ctx->stop_pool.push_back(f); }
    ctx->stop_queue.clear();
}

static bool is_pow2(uint64_t x) { return x && !(x & (x - 1)); }

static void journal_recycle(fgpu_ctx* ctx);
static int journal_add(fgpu_ctx* ctx, BatchBufs* b, const fgpu_reads* reads);
static int scan_replay(fgpu_ctx* ctx);

extern "C" {

int fgpu_abi_version(void) { return FGPU_ABI_VERSION; }

int fgpu_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int usable = 0;
    for (int d = 0; d < n; d++) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) == hipSuccess && std::string(prop.gcnArchName).rfind("gfx950", 0) == 0) usable++;
    }
    return usable;
}

const char* fgpu_last_error(const fgpu_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int fgpu_create(const fgpu_params* p, fgpu_ctx** out) {
    if (!p || !out) return FGPU_ERR_ARG;
    *out = nullptr;
    if (p->k < 1 || p->k > 31) { g_create_error = "k must be in 1..31"; return FGPU_ERR_ARG; }
    if (p->j < 0 || p->j > 8) { g_create_error = "j must be in 0..8"; return FGPU_ERR_ARG; }
    if (p->n_hash < 1 || p->n_hash > 10) { g_create_error = "n_hash must be in 1..10"; return FGPU_ERR_ARG; }
    if (!is_pow2(p->tai) || p->tai < 128) { g_create_error = "tai must be a power of two >= 128"; return FGPU_ERR_ARG; }
    if (p->max_spacer_dist < 1) { g_create_error = "max_spacer_dist must be >= 1"; return FGPU_ERR_ARG; }
    if (p->junction_capacity && !is_pow2(p->junction_capacity)) { g_create_error = "junction_capacity must be a power of two"; return FGPU_ERR_ARG; }
    // FGPU_CLI_TIMES=1: where the context's creation goes (stderr; the CLI's phase clock shows it as one line)
    const bool tell = getenv("FGPU_CLI_TIMES") != nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (tell) fprintf(stderr, "[fgpu_create] %-34s at %8.2f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count());
    };
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    lap("hipGetDeviceCount (runtime start)");
    if (e != hipSuccess || ndev == 0) {
        g_create_error = std::string("no HIP device: ") + (e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        return FGPU_ERR_HIP;
    }
    if (p->device < 0 || p->device >= ndev) { g_create_error = "device ordinal out of range"; return FGPU_ERR_ARG; }
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, p->device)) != hipSuccess) { g_create_error = hipGetErrorString(e); return FGPU_ERR_HIP; }
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
        g_create_error = std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
        return FGPU_ERR_HIP;
    }
    if ((e = hipSetDevice(p->device)) != hipSuccess) { g_create_error = hipGetErrorString(e); return FGPU_ERR_HIP; }
    lap("device properties, hipSetDevice");

    fgpu_ctx* ctx = new fgpu_ctx();
    ctx->prm = *p;
    // Beside the stream and buffer set-up below, a helper thread loads the code objects of the units the first batches use (the runtime
    // loads a unit at the first use of one of its kernels: 20-25 ms for the large ones) and creates the text stream (a queue: 10 ms).
    const int device = p->device;
    std::thread warm([device, ctx] {
        if (hipSetDevice(device) != hipSuccess) return;
        fgpu_touch_text();
        fgpu_touch_pack();
        fgpu_touch_load();
        (void)fgpu_text_streams(ctx);
        (void)hipGetLastError();
    });
    struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{warm};
    if (!ctx->prm.junction_capacity) {
        // default: one slot per 32 filter bits (2^24 slots for config 2's 2^29-bit filters, 2^27 for 2^32 bits), within 2^22..2^28;
        // at the reference's ~0.1 junctions per read that leaves the table below 10 % load
        uint64_t c = p->tai / 32;
        if (c < (1ULL << 22)) c = 1ULL << 22;
        if (c > (1ULL << 28)) c = 1ULL << 28;
        ctx->prm.junction_capacity = c;
    }
    if (!ctx->prm.max_batch_bases) ctx->prm.max_batch_bases = 1ULL << 30;
    // load batches kept in HBM for the scan of the same reads: 4 bits per base, at most an eighth of the device memory
    ctx->resident_budget = (p->flags & FGPU_FLAG_NO_RESIDENT) ? 0 : prop.totalGlobalMem / 8;
    ctx->journal_budget = prop.totalGlobalMem / 8;   // packed reads of a lazy scan kept for a replay (3 bits per base)
    if (const char* e = getenv("FGPU_JOURNAL_MB")) ctx->journal_budget = (uint64_t)atoll(e) << 20;   // tests: force the "outgrown" path
    ctx->fd.k = p->k;
    ctx->fd.j = p->j;
    ctx->fd.n_hash = p->n_hash;
    ctx->fd.max_spacer = p->max_spacer_dist;
    ctx->fd.kmask = (1ULL << (2 * p->k)) - 1;
    ctx->fd.tai_mask = p->tai - 1;
    ctx->profile = (p->flags & FGPU_FLAG_PROFILE) != 0;
    ctx->record_stops = (p->flags & FGPU_FLAG_RECORD_STOPS) != 0;
    { const char* e = getenv("FGPU_PROFILE_WALK"); ctx->prof_walk_detail = e && e[0] == '1'; }
    ctx->bloom_bytes = p->tai / 8;
    // Thin coverage per window -- a large genome, hence a large filter -- leaves clusters tiny however long the window: the bound grows
    // with the filter, 2^26 positions up to 2^30 filter bits, 2^28 from 2^32 on (config 5's walk stage 404 / 347 / 317 ms at 2^26 / 2^27 /
    // 2^28, config 4's 1 266 / 1 020 / 921 ms; config 2 settles at 2^25-2^26 whatever the bound).  The window table is 32 bytes per position
    // of the bound.
    ctx->max_span = std::min<uint64_t>(std::max<uint64_t>(p->tai >> 4, FGPU_MAX_SPAN), 1ULL << 28);
    if (const char* e = getenv("FGPU_MAX_SPAN_LOG2")) {   // experiment knob
        int l = atoi(e);
        if (l >= 12 && l <= 28) ctx->max_span = 1ULL << l;
    }
    memset(&ctx->load_stats, 0, sizeof(ctx->load_stats));
    memset(&ctx->scan_stats, 0, sizeof(ctx->scan_stats));
    int rc = FGPU_OK;
    auto fail = [&](const char* what, hipError_t he) {
        g_create_error = std::string(what) + ": " + hipGetErrorString(he);
        rc = FGPU_ERR_HIP;
    };
    if (p->stream) {
        ctx->stream = (hipStream_t)p->stream;
    } else {
        if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) fail("hipStreamCreate", e);
        ctx->own_stream = true;
    }
    ctx->launch_stream = ctx->stream;
    if (!rc) {
        // what only the scan needs is made beside pass 1 (fgpu_bg_join): the walk stream, the clean stream, their events, and the two largest
        // code objects -- 50-70 ms that used to sit between the command line and the first byte read ("What's weak" 4)
        ctx->bg = new std::thread([device, ctx] {
            auto bad = [ctx](const char* what, hipError_t he) {
                ctx->bg_err = std::string(what) + ": " + hipGetErrorString(he);
                ctx->bg_rc = FGPU_ERR_HIP;
            };
            hipError_t he;
            if ((he = hipSetDevice(device)) != hipSuccess) return bad("hipSetDevice", he);
            int lo = 0, hi = 0;
            hipDeviceGetStreamPriorityRange(&lo, &hi);   // hi = numerically lowest = highest priority
            if ((he = hipStreamCreateWithPriority(&ctx->wstream, hipStreamNonBlocking, hi)) != hipSuccess) return bad("hipStreamCreate (walk)", he);
            if ((he = hipStreamCreateWithPriority(&ctx->cstream, hipStreamNonBlocking, hi)) != hipSuccess) return bad("hipStreamCreate (clean)", he);
            if ((he = hipStreamCreateWithPriority(&ctx->ostream, hipStreamNonBlocking, hi)) != hipSuccess) return bad("hipStreamCreate (optimistic walk)", he);
            if ((he = hipEventCreateWithFlags(&ctx->ev_walked, hipEventDisableTiming)) != hipSuccess) return bad("hipEventCreate", he);
            if ((he = hipEventCreateWithFlags(&ctx->ev_listed, hipEventDisableTiming)) != hipSuccess) return bad("hipEventCreate", he);
            if ((he = hipEventCreateWithFlags(&ctx->ev_settled, hipEventDisableTiming)) != hipSuccess) return bad("hipEventCreate", he);
            for (int q = 0; q < 2; q++)
                if ((he = hipEventCreateWithFlags(&ctx->ev_uf_reset[q], hipEventDisableTiming)) != hipSuccess) return bad("hipEventCreate", he);
            fgpu_touch_scan_pure();
            fgpu_touch_scan_walk();
            (void)hipGetLastError();
        });
    }
    lap("main stream");
    // FGPU_DEBUG_ALLOC_FIRST=1 (measurement, scripts/kinds_probe.py): the first-set times -- the context's largest allocation, 4 bytes per filter
    // bit -- are asked for before anything else of the context instead of at the first fgpu_load_begin
    if (!rc && getenv("FGPU_DEBUG_ALLOC_FIRST") && hipMalloc(&ctx->first, p->tai * 4) != hipSuccess) { (void)hipGetLastError(); ctx->first = nullptr; }
    if (!rc && (e = hipMalloc(&ctx->bloo1, ctx->bloom_bytes)) != hipSuccess) fail("hipMalloc bloo1", e);
    if (!rc && (e = hipMalloc(&ctx->bloo2, ctx->bloom_bytes)) != hipSuccess) fail("hipMalloc bloo2", e);
    if (!rc && (e = hipMalloc(&ctx->counters, sizeof(DevCounters))) != hipSuccess) fail("hipMalloc counters", e);
    if (!rc && (e = hipHostMalloc(&ctx->counters_host, sizeof(DevCounters))) != hipSuccess) fail("hipHostMalloc", e);
    if (!rc && (e = hipHostMalloc(&ctx->fb_host, 64)) != hipSuccess) fail("hipHostMalloc", e);
    if (!rc) {
        memset(ctx->counters_host, 0, sizeof(DevCounters));
        hipMemsetAsync(ctx->bloo1, 0, ctx->bloom_bytes, ctx->stream);
        hipMemsetAsync(ctx->bloo2, 0, ctx->bloom_bytes, ctx->stream);
        hipMemsetAsync(ctx->counters, 0, sizeof(DevCounters), ctx->stream);
        if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) fail("initial memset", e);
    }
    lap("filters allocated and cleared");
    warm.join();   // (before the context can be destroyed)
    lap("pass-1 code objects, text stream");
    if (rc) {
        fgpu_destroy(ctx);
        return rc;
    }
    *out = ctx;
    return FGPU_OK;
}

void fgpu_destroy(fgpu_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->prm.device);      // (a host thread that drives several contexts: the calls below act on THIS context's device)
    (void)fgpu_bg_join(ctx);
    // nothing this context has queued is still running when its memory goes (round 6 audit, VERDICT r5 weak 1: the walk's side streams were
    // only waited for AFTER the buffers they work on had been freed -- hipFree waits for the device by itself, so no fault was ever seen)
    if (ctx->ostream) hipStreamSynchronize(ctx->ostream);
    if (ctx->cstream) hipStreamSynchronize(ctx->cstream);
    if (ctx->copy_stream) { hipStreamSynchronize(ctx->copy_stream); hipStreamDestroy(ctx->copy_stream); hipEventDestroy(ctx->copy_after); }
    if (ctx->tstream) {
        hipStreamSynchronize(ctx->tstream);
        hipStreamDestroy(ctx->tstream);
    }
    for (int i = 0; i < FGPU_TEXT_SETS; i++) if (ctx->ev_text_mark[i]) hipEventDestroy(ctx->ev_text_mark[i]);
    for (int i = 0; i < 2; i++) if (ctx->ev_stage_free[i]) hipEventDestroy(ctx->ev_stage_free[i]);
    if (ctx->ev_text_done) hipEventDestroy(ctx->ev_text_done);
    if (ctx->wstream) hipStreamSynchronize(ctx->wstream);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    for (PendingEvent& pe : ctx->pending_events) { hipEventDestroy(pe.a); hipEventDestroy(pe.b); }
    for (DevBuf* b : ctx->owned) if (b->p) hipFree(b->p);
    if (ctx->lp.flips_host) hipHostFree(ctx->lp.flips_host);
    if (ctx->lp.ev_open) hipEventDestroy(ctx->lp.ev_open);
    void* ptrs[] = {ctx->lp.bits, ctx->lp.first, ctx->short_pf, ctx->bloo1, ctx->bloo2, ctx->first, ctx->pair, ctx->rec, ctx->jkeys, ctx->jrecs, ctx->jstamps, ctx->jfilter, ctx->wkeys,
                    ctx->wbits, ctx->uf_parent, ctx->cl_count, ctx->cl_offset, ctx->cl_fill, ctx->cl_fail, ctx->ko_hk, ctx->ko_occ, ctx->ko_piece, ctx->cl_members, ctx->cl_roots, ctx->counters,
                    ctx->wdesc};
    for (void* p : ptrs) if (p) hipFree(p);
    stop_queue_recycle(ctx);
    for (StopBatch& f : ctx->stop_pool) hipHostFree(f.data);
    ctx->stop_pool.clear();
    if (ctx->counters_host) hipHostFree(ctx->counters_host);
    if (ctx->fb_host) hipHostFree(ctx->fb_host);
    if (ctx->wstream) { hipStreamSynchronize(ctx->wstream); hipStreamDestroy(ctx->wstream); }
    if (ctx->cstream) { hipStreamSynchronize(ctx->cstream); hipStreamDestroy(ctx->cstream); }
    if (ctx->ostream) { hipStreamSynchronize(ctx->ostream); hipStreamDestroy(ctx->ostream); }
    if (ctx->ev_walked) hipEventDestroy(ctx->ev_walked);
    if (ctx->ev_listed) hipEventDestroy(ctx->ev_listed);
    if (ctx->ev_settled) hipEventDestroy(ctx->ev_settled);
    for (int q = 0; q < 2; q++) if (ctx->ev_uf_reset[q]) hipEventDestroy(ctx->ev_uf_reset[q]);
    if (ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
    for (ResidentBatch* r : ctx->resident) delete r;
    for (JournalBatch* j : ctx->journal) delete j;
    for (JournalBatch* j : ctx->journal_pool) delete j;
    for (BatchBufs* b : ctx->all_batches) {
        if (b->pure_done) hipEventDestroy(b->pure_done);
        if (b->walk_done) hipEventDestroy(b->walk_done);
        delete b;
    }
    delete ctx;
}

int fgpu_synchronize(fgpu_ctx* ctx) {
    if (!ctx) return FGPU_ERR_ARG;
    return sync_all(ctx);
}

// ---- pass 1 --------------------------------------------------------------------------------------------------
int fgpu_load_begin(fgpu_ctx* ctx, int keep_carry) {
    if (!ctx) return FGPU_ERR_ARG;
    if (ctx->phase != 0) { ctx->err = "load_begin while another pass is open"; return FGPU_ERR_STATE; }
    FGPU_HIP(hipSetDevice(ctx->prm.device));
    if (int rc = fgpu_bloom_download_wait(ctx)) return rc;   // a download still in flight reads the filters this pass rewrites
    // Where the pass keeps its state (load.hip, Filt): the interleaved pair + first[], or -- FGPU_LOAD_LAYOUT=records: measured in round 5, a
    // faster marking kernel on filters of 2^32 bits and more but no faster pass, not the default -- 256-byte records.  Records need 8 bytes per
    // filter bit: where they cannot be had the pass falls back to the pair layout (4.25 bytes per bit).
    static const char* layout_env = getenv("FGPU_LOAD_LAYOUT");
    bool want_rec = layout_env && layout_env[0] == 'r';
    if (want_rec && !ctx->rec) {
        if (hipMalloc(&ctx->rec, ctx->prm.tai / 32 * 256) != hipSuccess) {
            (void)hipGetLastError();
            ctx->rec = nullptr;
            want_rec = false;
        }
    }
    ctx->rec_layout = want_rec;
    if (!ctx->rec_layout && !ctx->first) {
        hipError_t e = hipMalloc(&ctx->first, ctx->prm.tai * 4);
        if (e != hipSuccess) {
            ctx->err = std::string("hipMalloc of the first-set-time array (4 bytes per Bloom bit) failed: ") + hipGetErrorString(e);
            ctx->first = nullptr;
            return FGPU_ERR_NOMEM;
        }
    }
    if (!ctx->rec_layout && !ctx->pair) {   // working copy of both filters, interleaved word by word, for the duration of a load pass
        hipError_t e = hipMalloc(&ctx->pair, ctx->bloom_bytes * 2);
        if (e != hipSuccess) {
            ctx->err = std::string("hipMalloc of the interleaved filter pair failed: ") + hipGetErrorString(e);
            ctx->pair = nullptr;
            return FGPU_ERR_NOMEM;
        }
        if (int rc = fgpu_place_pair(ctx)) return rc;     // large filters: where the pair lies relative to first[] decides the marking kernel's speed (diag.hip)
    }
    // The carry of the following batches: brought up to date by sweeps of first[] (4 bytes per filter bit, streaming) that close epochs of
    // batches -- after batches 0, 1, 3, 7 ... the carry may lag, see fgpu_stage_load -- or, in between, by re-hashing a batch's new
    // k-mers right after it (k_carry_set: the carry never lags, one pass over the batch more).  Measured with both on the shapes that have
    // large filters (scripts/carry_mode_sweep.sh, round 3; per step, set / sweep): 2^31 bits, 10 M reads 185 / 192 ms; 2^32 bits, 10 M reads
    // 232 / 227; 2^33 bits: 25 M reads 581 / 554, config 5 (50 M x 150) 1 425 / 1 236, config 4 whole (200 M reads) 3 160 / 2 941 --
    // the longer the pass and the thinner the coverage of an epoch, the less an up-to-date carry is worth its pass.  So: sweeps,
    // except at 2^31 bits.  (One rank's passes of the 8-rank shape, 2^32 bits, same box, re-hashing / sweeping: the presence protocol's
    // load on a carried-in state 88-97 / 95 ms, the fix-up protocol's own load 146-163 / 143 ms.  Round 2 re-hashed from 2^31 bits up;
    // it had only measured short passes.)
    static const char* carry_env = getenv("FGPU_CARRY_MODE");   // "sweep" / "set": measurement aid
    ctx->carry_by_set = carry_env ? carry_env[0] == 's' && carry_env[1] == 'e' : ctx->prm.tai == (1ULL << 31);
    ctx->shard_times = (keep_carry & FGPU_LOAD_SHARD_TIMES) != 0;
    // ONE predicate for the fail planes: they exist for at most 4 hash functions (load.hip MISS_PLANES).  With more the flag is
    // accepted and the pass is a plain load -- nothing is sized for or copied from planes that are never made -- and fgpu_load_fixup answers
    // FGPU_ERR_STATE as faucet_gpu.h says (hosts then take the presence protocol: sharded.fixup_possible, shard_host.h)
    ctx->shard_planes = (keep_carry & FGPU_LOAD_SHARD_PLANES) != 0 && !ctx->shard_times && ctx->prm.n_hash <= 4;
    ctx->fixup_ready = false;
    ctx->pass_positions = ctx->pass_batches = 0;
    ctx->pass_empty_carry = !(keep_carry & FGPU_LOAD_KEEP_CARRY);
    keep_carry &= FGPU_LOAD_KEEP_CARRY;
    ctx->epoch_positions = ctx->swept_positions = 0;
    ctx->sweep_num = ctx->sweep_den = 1;
    const char* sweep_env = getenv("FGPU_SWEEP_RATIO");   // "num/den"; "0/1" = after every batch: measurement aid
    if (sweep_env) {
        unsigned n = 1, d = 1;
        if (sscanf(sweep_env, "%u/%u", &n, &d) == 2 && d > 0) { ctx->sweep_num = n; ctx->sweep_den = d; }
    }
    {   // a sweep streams tai x 4 bytes whatever the epoch holds: not before the epoch's accesses are tai / 16 (FGPU_SWEEP_MIN_FRAC; 0 = no such
        // bar, rounds 1-4).  The first sweeps of a pass on large filters came after a quarter-size ramp batch and brought the carry almost nothing:
        // one rank's own load of config 4 on 8 GPUs 350 -> 337 ms (6 sweeps -> 4), configs 2, 4, 5 whole unchanged (profiles/r05_sweep_policy.txt)
        const char* e = getenv("FGPU_SWEEP_MIN_FRAC");
        const double frac = e ? atof(e) : 16.0;
        ctx->sweep_min = frac > 0.0 ? (uint64_t)((double)ctx->prm.tai / frac / (double)std::max(1, ctx->prm.n_hash)) : 0;
    }
    if (!ctx->rec_layout) FGPU_HIP(hipMemsetAsync(ctx->first, 0xFF, ctx->prm.tai * 4, ctx->stream));   // (records: k_rec_init, fgpu_load_pair_begin)
    fgpu_resident_reset(ctx, true);
    if (!keep_carry) FGPU_HIP(hipMemsetAsync(ctx->bloo1, 0, ctx->bloom_bytes, ctx->stream));
    FGPU_HIP(hipMemsetAsync(ctx->bloo2, 0, ctx->bloom_bytes, ctx->stream));
    int rc = fgpu_load_pair_begin(ctx);
    if (rc) return rc;
    FGPU_HIP(hipMemsetAsync(ctx->counters, 0, sizeof(DevCounters), ctx->stream));
    memset(&ctx->load_stats, 0, sizeof(ctx->load_stats));
    ctx->host_waits = 0;
    ctx->host_wait_ms = 0;
    ctx->wait_sites.clear();
    ctx->phase = 1;
    return FGPU_OK;
}

static int check_reads(fgpu_ctx* ctx, const fgpu_reads* r) {
    if (!r || (r->n_reads && (!r->bases || !r->offsets))) { ctx->err = "null read batch"; return FGPU_ERR_ARG; }
    return FGPU_OK;
}

int fgpu_load_batch(fgpu_ctx* ctx, const fgpu_reads* reads) {
    if (!ctx) return FGPU_ERR_ARG;
    if (ctx->phase != 1) { ctx->err = "load_batch outside load_begin/load_end"; return FGPU_ERR_STATE; }
    int rc = check_reads(ctx, reads);
    if (rc) return rc;
    FGPU_HIP(hipSetDevice(ctx->prm.device));
    if ((rc = fgpu_stage_pack(ctx, reads))) return rc;
    if ((rc = fgpu_stage_load(ctx))) return rc;
    if (ctx->cur->T) ctx->pass_batches++;
    ctx->load_stats.reads_processed += reads->n_reads;
    return fgpu_host_batch_done(ctx, reads);   // (the host buffers were free again when the copy had run: fgpu_stage_pack)
}

int fgpu_presence_batch(fgpu_ctx* ctx, const fgpu_reads* reads) {
    if (!ctx) return FGPU_ERR_ARG;
    if (ctx->phase != 0) { ctx->err = "presence_batch while a pass is open"; return FGPU_ERR_STATE; }
    int rc = check_reads(ctx, reads);
    if (rc) return rc;
    FGPU_HIP(hipSetDevice(ctx->prm.device));
    if ((rc = fgpu_stage_pack(ctx, reads))) return rc;
    if ((rc = fgpu_stage_presence(ctx))) return rc;
    return fgpu_host_batch_done(ctx, reads);
}

int fgpu_load_end(fgpu_ctx* ctx, fgpu_load_stats* stats) {
    if (!ctx) return FGPU_ERR_ARG;
    if (ctx->phase != 1) { ctx->err = "load_end without load_begin"; return FGPU_ERR_STATE; }
    int rc = fgpu_load_sweep(ctx);          // bits set since the last sweep join bloo1
    if (!rc) rc = fgpu_load_pair_end(ctx);   // bloo1 / bloo2 back as the two raw bit arrays of the .bloom format
    if (!rc) rc = pull_counters(ctx);
    ctx->phase = 0;
    if (rc) return rc;
    ctx->load_stats.kmers = ctx->counters_host->kmers;
    ctx->load_stats.to_bloo2 = ctx->counters_host->to_bloo2;
    ctx->load_mark_hits = ctx->counters_host->mark_hits;
    ctx->load_mark_pending = ctx->counters_host->mark_pending;
    ctx->load_stats.unambiguous_reads = ctx->counters_host->segments;
    if (stats) *stats = ctx->load_stats;
    ctx->fixup_ready = (ctx->shard_times || ctx->shard_planes) && ctx->pass_empty_carry &&
                       ctx->resident_count == ctx->pass_batches && !(ctx->prm.flags & FGPU_FLAG_MERCY);
    return FGPU_OK;
}

int fgpu_load_fixup_state(fgpu_ctx* ctx, int* ready, uint64_t* resident_budget_bytes) {
    if (!ctx) return FGPU_ERR_ARG;
    if (ready) *ready = ctx->phase == 0 && ctx->fixup_ready ? 1 : 0;
    if (resident_budget_bytes) *resident_budget_bytes = ctx->resident_budget;
    return FGPU_OK;
}

int fgpu_load_fixup(fgpu_ctx* ctx, const void* prefix_dev, fgpu_load_stats* stats) {
    if (!ctx || !prefix_dev) return FGPU_ERR_ARG;
    if (ctx->phase != 0) { ctx->err = "load_fixup while a pass is open"; return FGPU_ERR_STATE; }
    if (!ctx->fixup_ready) {
        ctx->err = "load_fixup needs a finished load pass begun with FGPU_LOAD_SHARD_TIMES and an empty carry (or FGPU_LOAD_SHARD_PLANES and at most 4 hash functions), every batch kept resident, no --mercy";
        return FGPU_ERR_STATE;
    }
    FGPU_HIP(hipSetDevice(ctx->prm.device));
    if (int rc = fgpu_bloom_download_wait(ctx)) return rc;
    int rc = fgpu_stage_fixup(ctx, (const uint32_t*)prefix_dev);
    if (!rc) rc = fgpu_util_or(ctx, ctx->bloo1, prefix_dev, ctx->bloom_bytes);
    if (!rc) rc = pull_counters(ctx);
    if (rc) return rc;
    ctx->fixup_ready = false;                       // once per pass: the planes now speak about the global filter
    ctx->load_stats.to_bloo2 = ctx->counters_host->to_bloo2;
    if (stats) *stats = ctx->load_stats;
    return FGPU_OK;
}

static uint32_t* bloom_ptr(fgpu_ctx* ctx, int which) { return which == FGPU_BLOO1 ? ctx->bloo1 : which == FGPU_BLOO2 ? ctx->bloo2 : nullptr; }

int fgpu_bloom_download(fgpu_ctx* ctx, int which, uint8_t* host_out, uint64_t nbytes) {
    if (!ctx || !host_out || !bloom_ptr(ctx, which) || nbytes != ctx->bloom_bytes) return FGPU_ERR_ARG;
    if (ctx->phase == 1) { ctx->err = "bloom_download inside a load pass: the filters are interleaved until load_end"; return FGPU_ERR_STATE; }
    FGPU_HIP(hipMemcpyAsync(host_out, bloom_ptr(ctx, which), nbytes, hipMemcpyDeviceToHost, ctx->stream));
    FGPU_HIP(fgpu_sync_stream(ctx, ctx->stream));
    return FGPU_OK;
}

int fgpu_bloom_download_wait(fgpu_ctx* ctx) {
    if (!ctx) return FGPU_ERR_ARG;
    if (!ctx->copy_pending) return FGPU_OK;
    ctx->copy_pending = false;
    FGPU_HIP(fgpu_sync_stream(ctx, ctx->copy_stream));
    return FGPU_OK;
}

int fgpu_bloom_download_begin(fgpu_ctx* ctx, int which, uint8_t* host_out, uint64_t nbytes) {
    if (!ctx || !host_out || !bloom_ptr(ctx, which) || nbytes != ctx->bloom_bytes) return FGPU_ERR_ARG;
    if (ctx->phase == 1) { ctx->err = "bloom_download_begin inside a load pass: the filters are interleaved until load_end"; return FGPU_ERR_STATE; }
    if (int rc = fgpu_bloom_download_wait(ctx)) return rc;
    FGPU_HIP(hipSetDevice(ctx->prm.device));
    if (!ctx->copy_stream) {
        FGPU_HIP(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        FGPU_HIP(hipEventCreateWithFlags(&ctx->copy_after, hipEventDisableTiming));
    }
    FGPU_HIP(hipEventRecord(ctx->copy_after, ctx->stream));
    FGPU_HIP(hipStreamWaitEvent(ctx->copy_stream, ctx->copy_after, 0));
    FGPU_HIP(hipMemcpyAsync(host_out, bloom_ptr(ctx, which), nbytes, hipMemcpyDeviceToHost, ctx->copy_stream));
    ctx->copy_pending = true;
    return FGPU_OK;
}

int fgpu_bloom_upload(fgpu_ctx* ctx, int which, const uint8_t* host_in, uint64_t nbytes) {
    if (!ctx || !host_in || !bloom_ptr(ctx, which) || nbytes != ctx->bloom_bytes) return FGPU_ERR_ARG;
    if (ctx->phase != 0) { ctx->err = "bloom_upload while a pass is open (a load pass would overwrite it at load_end, a scan is reading bloo2)"; return FGPU_ERR_STATE; }
    if (int rc = fgpu_bloom_download_wait(ctx)) return rc;
    if (which == FGPU_BLOO2) fgpu_resident_reset(ctx, false);   // the kept "routed to bloo2" planes speak about the filter this replaces
    FGPU_HIP(hipMemcpyAsync(bloom_ptr(ctx, which), host_in, nbytes, hipMemcpyHostToDevice, ctx->stream));
    FGPU_HIP(fgpu_sync_stream(ctx, ctx->stream));
    return FGPU_OK;
}

int fgpu_bloom_weight(fgpu_ctx* ctx, int which, float* weight) {
    if (!ctx || !weight || !bloom_ptr(ctx, which)) return FGPU_ERR_ARG;
    if (ctx->phase == 1) { ctx->err = "bloom_weight inside a load pass: the filters are interleaved until load_end"; return FGPU_ERR_STATE; }
    FGPU_HIP(hipMemsetAsync(&ctx->counters->pad, 0, 8, ctx->stream));
    int rc = fgpu_util_popcount(ctx, bloom_ptr(ctx, which), ctx->bloom_bytes, &ctx->counters->pad);
    if (rc) return rc;
    FGPU_HIP(hipMemcpyAsync(&ctx->counters_host->pad, &ctx->counters->pad, 8, hipMemcpyDeviceToHost, ctx->stream));
    FGPU_HIP(fgpu_sync_stream(ctx, ctx->stream));
    *weight = (float)(long)ctx->counters_host->pad / (float)ctx->prm.tai;   // Bloom::weight: (float)weight/(float)tai
    return FGPU_OK;
}

int fgpu_bloom_devptr(fgpu_ctx* ctx, int which, void** dptr, uint64_t* nbytes) {
    if (!ctx || !dptr || !bloom_ptr(ctx, which)) return FGPU_ERR_ARG;
    *dptr = bloom_ptr(ctx, which);
    if (nbytes) *nbytes = ctx->bloom_bytes;
    return FGPU_OK;
}

int fgpu_bitmap_or(fgpu_ctx* ctx, void* dst_dev, const void* src_dev, uint64_t nbytes) {
    if (!ctx || !dst_dev || !src_dev || (nbytes & 15)) return FGPU_ERR_ARG;
    return fgpu_util_or(ctx, dst_dev, src_dev, nbytes);
}

// ---- pass 2 --------------------------------------------------------------------------------------------------
int fgpu_scan_begin(fgpu_ctx* ctx) {
    if (!ctx) return FGPU_ERR_ARG;
    if (ctx->phase != 0) { ctx->err = "scan_begin while another pass is open"; return FGPU_ERR_STATE; }
    FGPU_HIP(hipSetDevice(ctx->prm.device));
    int rc = fgpu_bg_join(ctx);      // the walk's streams and events (made beside pass 1, fgpu_create)
    if (rc) return rc;
    if ((rc = fgpu_scan_alloc(ctx))) return rc;
    if ((rc = fgpu_scan_reset(ctx))) return rc;
    ctx->fixup_ready = false;   // the load pass' counters go with this reset
    FGPU_HIP(hipMemsetAsync(ctx->counters, 0, sizeof(DevCounters), ctx->stream));
    memset(&ctx->scan_stats, 0, sizeof(ctx->scan_stats));
    memset(ctx->counters_host, 0, sizeof(DevCounters));
    ctx->scan_windows = 0;
    ctx->scan_pieces_seen = 0;
    ctx->scan_piece_base = 0;
    ctx->scan_imported = 0;
    ctx->delta_next = 0;
    journal_recycle(ctx);
    ctx->journal_on = !(ctx->prm.flags & FGPU_FLAG_EAGER_FLAGS) && !ctx->eager_runtime;
    ctx->eager_scan = ctx->lazy_failed = ctx->capacity_failed = false;
    ctx->dl_keys_n = 0;
    ctx->late_acc[0] = ctx->late_acc[1] = ctx->late_acc[2] = 0;
    ctx->stops_delivered = 0;
    ctx->lp_applied_seq = 0;
    ctx->host_waits = 0;
    ctx->host_wait_ms = 0;
    ctx->wait_sites.clear();
    if (ctx->short_pf) FGPU_HIP(hipMemsetAsync(ctx->short_pf, 0, ctx->short_pf_tai / 8, ctx->stream));   // a scan starts with empty pair filters
    if ((rc = fgpu_long_pairs_reset(ctx))) return rc;
    ctx->have_import = false;
    ctx->journal_max_read_len = 0;
    ctx->hint_in_table = false;
    ctx->hint_gen = 0;
    ctx->hint_n = ctx->hint_max_seq = 0;
    ctx->delta_ready = false;
    ctx->delta_keys = ctx->refresh_full = ctx->refresh_delta = ctx->refresh_mismatch = 0;
    ctx->sparse_links = ctx->full_links = 0;
    ctx->dbg_stall_us = getenv("FGPU_DEBUG_WALK_STALL_US") ? std::min(100000, std::max(0, atoi(getenv("FGPU_DEBUG_WALK_STALL_US")))) : 0;
    ctx->dbg_delta_check = getenv("FGPU_DEBUG_DELTA_CHECK") != nullptr;
    ctx->no_sparse_link = getenv("FGPU_NO_SPARSE_LINK") != nullptr;      // measurement aid: no candidate planes, every window of a prepared batch linked in full
    // calibrated upwards window by window; a context that has scanned before starts a quarter below where that scan ended up
    const uint64_t start_span = std::max<uint64_t>(1ULL << 18, ctx->settled_span / 4);
    ctx->window_span = ctx->prm.walk_window_span ? std::min<uint64_t>(std::max<uint64_t>(ctx->prm.walk_window_span, 64), ctx->max_span)
                                                 : std::min<uint64_t>(start_span, ctx->max_span);
    ctx->calib_left = 16;
    ctx->calib_f = ctx->calib_p = 0;
    ctx->adapt_followers = 0;
    ctx->adapt_pieces = 0;
    ctx->adapt_overflows = 0;
    ctx->adapt_vote = 0;
    ctx->calib_ovf = 0;
    ctx->span_ceiling = ~0ULL;
    ctx->repeats_seen_before = false;
    ctx->walked_pieces = 0;
    ctx->scan_batch_index = 0;
    ctx->scan_batch_seq = 0;
    for (BatchBufs* b : ctx->to_harvest) b->stops_pending = false;   // lists of an earlier scan nobody asked for
    ctx->to_harvest.clear();
    stop_queue_recycle(ctx);
    for (BatchBufs* b : ctx->prepared) ctx->pool.push_back(b);
    ctx->prepared.clear();
    ctx->cur = &ctx->bb_default;
    ctx->phase = 2;
    return FGPU_OK;
}

// Adapt the scheduling window to the data.  Larger windows mean fewer launches and fuller kernels but longer
// dependency chains inside the clusters; measured on config 2 the step time keeps falling until about a third of the
// pieces queue behind an earlier piece of their cluster (2^21: 254 ms, 2^22: 225, 2^23: 213, 2^24: 207, 2^25: 205).
// counters_host must be fresh with respect to the walks issued so far.
static void adapt_window(fgpu_ctx* ctx) {
    // Round 4.  On config 2 the share of queueing pieces is 0.25 / 0.43 / 0.58 at 2^24 / 2^25 / 2^26 positions and one batch's counters are a
    // noisy sample of it, so the controller of rounds 1-3 moved the size up and down between batches for good -- and every move asked the
    // host to wait for the next windows one by one, which keeps it from feeding the next batch's pure stage: 12 waits per step
    // (scripts/timeline.py), 121.4-122.1 ms a step against 116.8-118.9 with any fixed size in that range (scripts/span_fixed_ab.sh).
    // Now: both counts come from one snapshot of the device's counters (pieces counted behind their window's walks); growing on a share that
    // is not clear-cut (0.15-1/3) has to be asked for by two batches running; growing waits for windows only beyond the largest size a batch
    // of this context has been walked at, shrinking only towards a size it has not walked yet or when most pieces queue (the percolation
    // case the waits are for).  The thresholds themselves are unchanged, and shrinking still acts at once: a wider dead band, decisions held
    // back for a batch or two (shrinking included) and x4 steps were tried as well -- they walked config 4 at larger windows than suit it
    // and config 3's repeats at a percolating size now and then (pass 2 of 170 ms at 310-360).
    static const bool dbg_span = getenv("FGPU_DEBUG_SPAN") != nullptr;
    const uint64_t f = ctx->counters_host->followers_seen - ctx->adapt_followers;
    const uint64_t p = ctx->counters_host->walked_pieces - ctx->adapt_pieces;
    if (p == 0 || ctx->prm.walk_window_span) return;      // (no window has been completed since the last look at the counters)
    {
        const uint64_t usual = std::min<uint64_t>(std::min<uint64_t>(ctx->max_span, FGPU_USUAL_SPAN), ctx->span_ceiling);
        int want = 0;             // -1 smaller, +1 larger
        bool clear = false;
        if (f * 2 > p) { want = -1; clear = f * 5 > p * 4; }
        else if (f * 3 < p) { want = 1; clear = f * 20 < p * 3; }
        // too large is dear (a batch at a percolating size), too small is not: shrinking acts at once, growing on a clear share or the second time
        const bool act = want < 0 || (want > 0 && (clear || ctx->adapt_vote > 0));
        ctx->adapt_vote = act ? 0 : want;
        if (want >= 0) ctx->proven_span = std::max(ctx->proven_span, ctx->window_span);
        if (ctx->counters_host->ko_overflows > ctx->adapt_overflows && ctx->window_span > 4096) {
            // a window's large clusters outgrew the tables of the large-cluster walks and were walked piece after piece by their one thread:
            // far too large a window for this data
            ctx->window_span = std::max<uint64_t>(4096, ctx->window_span / 4);
            ctx->proven_span = std::min(ctx->proven_span, ctx->window_span);
            ctx->span_ceiling = std::min(ctx->span_ceiling, ctx->window_span * 2);     // (this scan does not grow to that size again)
            ctx->calib_ovf = ctx->counters_host->ko_overflows;
            ctx->calib_left = 0;
        }
        else if (act && want < 0 && ctx->window_span > 4096) {
            ctx->window_span /= 2;
            // ... and look again window by window -- unless this context has walked a batch at the smaller size before and the share is not
            // the percolation case (most pieces queueing) the looks are for
            if (clear || ctx->window_span > ctx->proven_span) ctx->calib_left = 8;
            ctx->proven_span = std::min(ctx->proven_span, ctx->window_span);
        }
        else if (act && want > 0 && ctx->window_span < usual) {
            // clusters percolate at a sharp threshold (about one genome coverage per window): a whole batch at a size that turns out to be
            // beyond it costs seconds, so the first windows at a size this context has not walked yet are looked at one by one
            ctx->window_span = std::min<uint64_t>(ctx->window_span * 2, usual);
            if (ctx->window_span > ctx->proven_span) ctx->calib_left = std::max(ctx->calib_left, 2);
        }
        else if (f * 16 < p && ctx->window_span < std::min<uint64_t>(ctx->max_span, ctx->span_ceiling)) ctx->window_span *= 2;   // thin coverage per window: see FGPU_MAX_SPAN
    }
    if (dbg_span) fprintf(stderr, "[span] batch: followers %llu of %llu pieces -> span %llu, looks %d, vote %d\n", (unsigned long long)f, (unsigned long long)p,
                          (unsigned long long)ctx->window_span, ctx->calib_left, ctx->adapt_vote);
    ctx->adapt_followers = ctx->counters_host->followers_seen;
    ctx->adapt_pieces = ctx->counters_host->walked_pieces;
    ctx->adapt_overflows = ctx->counters_host->ko_overflows;
}

static BatchBufs* acquire_batch(fgpu_ctx* ctx) {
    // FIFO over (at least) two BatchBufs, so that the one handed out was last walked two batches ago
    BatchBufs* b;
    // (at most FGPU_DELTA_RING: a batch registers the created-key lists of the FGPU_DELTA_RING - 1 batches before it)
    static const size_t depth = getenv("FGPU_SCAN_BUFFERS") ? (size_t)std::min(FGPU_DELTA_RING, std::max(2, atoi(getenv("FGPU_SCAN_BUFFERS")))) : 2;
    // The buffers handed out are the ones used `depth` batches ago -- NOT the oldest of the pool: a context that has walked a prepared shard
    // keeps dozens of buffers, and a streaming scan that took them first in, first out let its pure stage run as far ahead of the walk as the
    // pool is deep (nothing left to wait for), beyond what the created-key lists cover: keys missing from the snapshot planes, a wrong map
    // (seen at config 4's size, 50 M reads streamed after a 60 M-read prepared shard: 9 887 records too many).  The streaming scan pushes every
    // batch's buffers back at the end of the pool, so the one used `depth` batches ago sits `depth` from the end.
    if (ctx->pool.size() >= depth) {
        const size_t at = ctx->pool.size() - depth;
        b = ctx->pool[at];
        ctx->pool.erase(ctx->pool.begin() + at);
    } else {
        b = new BatchBufs();
        ctx->all_batches.push_back(b);
        hipEventCreateWithFlags(&b->pure_done, hipEventDisableTiming);
        hipEventCreateWithFlags(&b->walk_done, hipEventDisableTiming);
    }
    return b;
}

// after the walk of batch b has been issued: its lists become available (FGPU_FLAG_RECORD_STOPS)
static void note_walked(fgpu_ctx* ctx, BatchBufs* b) {
    if (!ctx->record_stops) return;
    b->stops_pending = true;
    ctx->to_harvest.push_back(b);
}

static int scan_pure_into(fgpu_ctx* ctx, BatchBufs* b, const fgpu_reads* reads) {
    if (b->stops_pending) {   // the buffers still hold the visit planes of an earlier batch: bring its lists to the host first
        int hrc = fgpu_scan_harvest(ctx, b);
        if (hrc) return hrc;      // (FGPU_INTERNAL_REPLAY: that batch's walk went wrong; the caller replays and comes back)
    }
    b->seq = ctx->scan_batch_seq++;
    ctx->cur = b;
    const double t_a = fgpu_host_now();
    if (b->walk_pending) {   // the walk stream may still be reading this batch's planes
        FGPU_HIP(fgpu_sync_event(ctx, b->walk_done));
        b->walk_pending = false;
    }
    const double t_b = fgpu_host_now();
    ctx->host_ms[0] += t_b - t_a;
    int rc = fgpu_stage_pack(ctx, reads);
    ctx->host_ms[1] += fgpu_host_now() - t_b;
    if (!rc) rc = journal_add(ctx, b, reads);
    uint64_t n_pieces = 0;
    if (!rc) rc = fgpu_stage_scan_pure(ctx, &n_pieces);   // ends with the batch's only synchronisation (piece count)
    if (!rc) ctx->journal_max_read_len = std::max<uint64_t>(ctx->journal_max_read_len, ctx->counters_host->max_read_len);
    if (!rc) rc = check_errors(ctx);
    if (!rc) ctx->scan_stats.reads_processed += reads->n_reads;
    if (!rc && b->pure_done) FGPU_HIP(hipEventRecord(b->pure_done, ctx->stream));
    if (!rc) rc = fgpu_host_batch_done(ctx, reads);
    return rc;
}

// ---- the scan's journal: lazy junction tests without a caller-visible fall-back ---------------------------------------------------------
// While a scan evaluates its junction tests lazily the packed form of every batch (codes + bad plane, 3 bits per base; the read offsets when
// scanInputRead's lists are recorded) stays in HBM, up to an eighth of the device memory.  If the walk meets a preview it cannot repair
// (never seen on real input, forced by FGPU_DEBUG_LAZY_FAIL=1) the library itself resets the junction map and scans the
// journal again with every test evaluated: same results, nothing for the caller to do -- ReadScanner::scanReads has no such contract
// either (src/ReadScanner.cpp:284-359).  A scan that outgrows the journal is first brought to a point where everything walked so far is
// known to be good, then goes on with eager tests (which cannot fail that way) and without a journal.
static void journal_recycle(fgpu_ctx* ctx) {
    for (JournalBatch* j : ctx->journal) ctx->journal_pool.push_back(j);
    ctx->journal.clear();
    ctx->journal_bytes = 0;
}

static int journal_add(fgpu_ctx* ctx, BatchBufs* b, const fgpu_reads* reads) {
    if (!ctx->journal_on || ctx->in_replay) return FGPU_OK;
    const uint64_t cb = 2 * (b->n_words + FGPU_PADW) * 8, bbytes = (b->n_words + FGPU_PADW) * 8, ob = ctx->record_stops ? (b->n_reads + 1) * 8 : 0;
    if (ctx->journal_bytes + cb + bbytes + ob > ctx->journal_budget) {
        // no room: everything walked so far is checked (and scanned again if need be), the rest of the scan is eager
        int rc = pull_counters(ctx);
        if (rc) return rc;
        BatchBufs* const cur = ctx->cur;
        // the batch being packed has already been folded into the device's longest-read maximum (fgpu_stage_pack) and is not in the journal:
        // the replay restores that maximum from journal_max_read_len, so it has to know this batch's reads too
        ctx->journal_max_read_len = std::max<uint64_t>(ctx->journal_max_read_len, ctx->counters_host->max_read_len);
        if (ctx->lazy_failed && (rc = scan_replay(ctx))) return rc;
        ctx->cur = cur;
        ctx->journal_on = false;
        ctx->eager_scan = true;
        journal_recycle(ctx);
        return FGPU_OK;
    }
    JournalBatch* j;
    if (!ctx->journal_pool.empty()) { j = ctx->journal_pool.back(); ctx->journal_pool.pop_back(); }
    else j = new JournalBatch();
    int rc;
    if ((rc = fgpu_ensure_b(ctx, &j->codes, cb)) || (rc = fgpu_ensure_b(ctx, &j->bad, bbytes)) || (ob && (rc = fgpu_ensure_b(ctx, &j->offs, ob)))) {
        ctx->journal_pool.push_back(j);
        return rc;
    }
    if (b->T) {
        FGPU_HIP(hipMemcpyAsync(j->codes.p, b->codes.p, cb, hipMemcpyDeviceToDevice, ctx->stream));
        FGPU_HIP(hipMemcpyAsync(j->bad.p, b->bad.p, bbytes, hipMemcpyDeviceToDevice, ctx->stream));
        if (ob) FGPU_HIP(hipMemcpyAsync(j->offs.p, b->d_offs, ob, hipMemcpyDeviceToDevice, ctx->stream));
    }
    j->T = b->T; j->n_words = b->n_words; j->n_reads = b->n_reads; j->seq = b->seq;
    ctx->journal.push_back(j);
    ctx->journal_bytes += cb + bbytes + ob;
    (void)reads;
    return FGPU_OK;
}

// Reset the junction map and scan every journalled batch again with all junction tests evaluated.  On return everything the journal holds
// has been walked; batches that were only prepared are walked as well (their turn has come: this is only reached from a walk or after one).
static int scan_replay_once(fgpu_ctx* ctx);
static int scan_replay(fgpu_ctx* ctx) {
    for (int attempt = 0;; attempt++) {
        const bool for_room = ctx->capacity_failed;
        ctx->capacity_failed = false;
        if (for_room) {
            int rc = sync_all(ctx);
            if (!rc) rc = fgpu_scan_regrow_empty(ctx, ctx->jcap * 4);
            if (rc) { ctx->lazy_failed = false; return rc; }
            ctx->capacity_replays++;
        }
        int rc = scan_replay_once(ctx);
        // the replay itself ran out of room (its batches are walked with the table kept below a quarter full between them, so this takes a
        // single batch that creates more than three times what the whole table held): once more, larger again
        if (rc == FGPU_ERR_CAPACITY && (ctx->counters_host->error_flags & 1ULL) && ctx->jcap < (1ULL << 31) && attempt < 6) {
            ctx->capacity_failed = true;
            continue;
        }
        // The journal still holds every batch of the scan: it stays on, so that a later batch that outgrows the table is absorbed as well (the
        // scan is eager from here on and cannot fail the lazy way again)
        if (!rc && for_room) ctx->journal_on = true;
        return rc;
    }
}

static int scan_replay_once(fgpu_ctx* ctx) {
    ctx->in_replay = true;
    ctx->journal_on = false;
    ctx->eager_scan = true;
    ctx->lazy_failed = false;
    int rc = sync_all(ctx);
    for (BatchBufs* b : ctx->prepared) ctx->pool.push_back(b);
    ctx->prepared.clear();
    for (BatchBufs* b : ctx->to_harvest) b->stops_pending = false;
    ctx->to_harvest.clear();
    stop_queue_recycle(ctx);
    if (!rc) rc = fgpu_scan_reset(ctx);
    if (rc) { ctx->in_replay = false; return rc; }
    {   // what fgpu_diag_late_flags reports of the voided attempt (everything has completed: sync_all above)
        unsigned long long ln[3] = {0, 0, 0};
        if (hipMemcpy(ln, ctx->counters->late_n, sizeof(ln), hipMemcpyDeviceToHost) == hipSuccess)
            for (int i = 0; i < 3; i++) ctx->late_acc[i] += ln[i];
    }
    FGPU_HIP(hipMemsetAsync(ctx->counters, 0, sizeof(DevCounters), ctx->stream));
    FGPU_HIP(hipMemcpyAsync(&ctx->counters->max_read_len, &ctx->journal_max_read_len, 8, hipMemcpyHostToDevice, ctx->stream));
    FGPU_HIP(fgpu_sync_stream(ctx, ctx->stream));
    const uint64_t reads_processed = ctx->scan_stats.reads_processed;
    memset(&ctx->scan_stats, 0, sizeof(ctx->scan_stats));
    ctx->scan_stats.reads_processed = reads_processed;
    memset(ctx->counters_host, 0, sizeof(DevCounters));
    ctx->scan_windows = 0;
    ctx->scan_pieces_seen = 0;
    ctx->scan_piece_base = 0;
    ctx->scan_imported = 0;
    ctx->delta_next = 0;
    ctx->hint_in_table = false;
    ctx->delta_ready = false;            // (the replay makes every batch's planes from scratch against the table as it stands)
    ctx->window_span = ctx->prm.walk_window_span ? std::min<uint64_t>(std::max<uint64_t>(ctx->prm.walk_window_span, 64), ctx->max_span)
                                                 : std::min<uint64_t>(std::max<uint64_t>(1ULL << 18, ctx->settled_span / 4), ctx->max_span);
    ctx->calib_left = 16;
    ctx->calib_f = ctx->calib_p = 0;
    ctx->adapt_followers = ctx->adapt_pieces = ctx->adapt_overflows = 0;
    ctx->adapt_vote = 0;
    ctx->calib_ovf = 0;
    ctx->span_ceiling = ~0ULL;
    ctx->repeats_seen_before = false;
    ctx->walked_pieces = 0;
    ctx->scan_batch_index = 0;
    memset(&ctx->carried, 0, sizeof(ctx->carried));
    if (ctx->have_import) {   // the table the previous shard handed over comes first again
        if ((rc = fgpu_scan_reserve(ctx, ctx->import_n)) || (rc = fgpu_scan_import_impl(ctx, ctx->import_copy.p, ctx->import_n))) { ctx->in_replay = false; return rc; }
        FGPU_HIP(fgpu_sync_stream(ctx, ctx->stream));
        ctx->scan_imported = ctx->import_n;
        if (ctx->import_has_carried) {
            ctx->carried = ctx->import_carried;
            ctx->scan_piece_base = ctx->import_carried.reads_no_errors;
        }
    }
    for (size_t i = 0; i < ctx->journal.size() && !rc; i++) {
        JournalBatch* j = ctx->journal[i];
        BatchBufs* b = acquire_batch(ctx);
        if (b->stops_pending && (rc = fgpu_scan_harvest(ctx, b))) break;   // lists of the replayed batch these buffers held two batches ago
        if (b->walk_pending) {
            FGPU_HIP(fgpu_sync_event(ctx, b->walk_done));
            b->walk_pending = false;
        }
        b->seq = j->seq;
        ctx->cur = b;
        b->T = j->T; b->n_words = j->n_words; b->n_reads = j->n_reads;
        b->d_offs = (const uint64_t*)j->offs.p;
        if (b->T) {
            const uint64_t cb = 2 * (b->n_words + FGPU_PADW) * 8, bbytes = (b->n_words + FGPU_PADW) * 8;
            if ((rc = fgpu_ensure_b(ctx, &b->codes, cb)) || (rc = fgpu_ensure_b(ctx, &b->bad, bbytes))) break;
            FGPU_HIP(hipMemcpyAsync(b->codes.p, j->codes.p, cb, hipMemcpyDeviceToDevice, ctx->stream));
            FGPU_HIP(hipMemcpyAsync(b->bad.p, j->bad.p, bbytes, hipMemcpyDeviceToDevice, ctx->stream));
        }
        uint64_t n_pieces = 0;
        rc = fgpu_stage_scan_pure(ctx, &n_pieces);
        if (!rc) rc = check_errors(ctx);
        if (!rc && b->pure_done) FGPU_HIP(hipEventRecord(b->pure_done, ctx->stream));
        if (!rc) rc = fgpu_scan_reserve(ctx, ctx->counters_host->n_junctions + ctx->scan_imported);
        if (!rc) {
            adapt_window(ctx);
            rc = fgpu_stage_scan_walk(ctx, b->n_pieces);
            ctx->walked_pieces += b->n_pieces;
            if (!rc) note_walked(ctx, b);
        }
        ctx->pool.push_back(b);
    }
    ctx->cur = &ctx->bb_default;
    if (!rc) rc = pull_counters(ctx);     // (still in_replay: an overflow of the last walks is reported here, to scan_replay, not flagged for later)
    ctx->in_replay = false;
    ctx->scan_replays++;
    return rc;
}

int fgpu_scan_batch(fgpu_ctx* ctx, const fgpu_reads* reads) {
    if (!ctx) return FGPU_ERR_ARG;
    if (ctx->phase != 2) { ctx->err = "scan_batch outside scan_begin/scan_end"; return FGPU_ERR_STATE; }
    if (!ctx->prepared.empty()) { ctx->err = "scan_batch while prepared batches are waiting: call fgpu_scan_walk_prepared first"; return FGPU_ERR_STATE; }
    if (ctx->hint_in_table) { ctx->err = "the junction table holds a preview (fgpu_scan_import_hint): import the real table before walking"; return FGPU_ERR_STATE; }
    int rc = check_reads(ctx, reads);
    if (rc) return rc;
    FGPU_HIP(hipSetDevice(ctx->prm.device));
    if (ctx->lazy_failed && (rc = scan_replay(ctx))) return rc;      // an earlier walk met a preview it could not repair: scan again, eagerly
    BatchBufs* b = acquire_batch(ctx);
    rc = scan_pure_into(ctx, b, reads);          // main stream; overlaps the previous batch's walk on the walk stream
    if (rc == FGPU_INTERNAL_REPLAY) {            // noticed while the buffers' previous lists were fetched: replay, then this batch from the start
        ctx->cur = &ctx->bb_default;
        ctx->pool.insert(ctx->pool.begin(), b);
        if ((rc = scan_replay(ctx))) return rc;
        b = acquire_batch(ctx);
        rc = scan_pure_into(ctx, b, reads);
    }
    if (!rc && ctx->lazy_failed) {
        // noticed at this batch's own synchronisation: the batch is in the journal (prepared lazily); the replay scans it with the others
        ctx->cur = &ctx->bb_default;
        ctx->pool.push_back(b);
        return scan_replay(ctx);
    }
    // records as of the pure stage's synchronisation (the previous walk may still be adding some): room for this batch's
    if (!rc) rc = fgpu_scan_reserve(ctx, ctx->counters_host->n_junctions + ctx->scan_imported);
    if (!rc) {
        adapt_window(ctx);                       // counters as of the pure stage's synchronisation (the walk may lag one batch)
        const double t_w = fgpu_host_now();
        rc = fgpu_stage_scan_walk(ctx, b->n_pieces);
        ctx->host_ms[5] += fgpu_host_now() - t_w;
        ctx->walked_pieces += b->n_pieces;
        if (!rc) note_walked(ctx, b);
    }
    ctx->cur = &ctx->bb_default;
    ctx->pool.push_back(b);
    return rc;
}

int fgpu_scan_prepare(fgpu_ctx* ctx, const fgpu_reads* reads) {
    if (!ctx) return FGPU_ERR_ARG;
    if (ctx->phase != 2) { ctx->err = "scan_prepare outside scan_begin/scan_end"; return FGPU_ERR_STATE; }
    int rc = check_reads(ctx, reads);
    if (rc) return rc;
    FGPU_HIP(hipSetDevice(ctx->prm.device));
    BatchBufs* b;
    if (!ctx->pool.empty()) { b = ctx->pool.front(); ctx->pool.erase(ctx->pool.begin()); }
    else {
        b = new BatchBufs();
        ctx->all_batches.push_back(b);
        hipEventCreateWithFlags(&b->pure_done, hipEventDisableTiming);
        hipEventCreateWithFlags(&b->walk_done, hipEventDisableTiming);
    }
    rc = scan_pure_into(ctx, b, reads);
    ctx->cur = &ctx->bb_default;
    if (rc == FGPU_INTERNAL_REPLAY) {   // (only after walks of this scan: prepare-only scans never get here)
        ctx->pool.insert(ctx->pool.begin(), b);
        if ((rc = scan_replay(ctx))) return rc;
        return fgpu_scan_prepare(ctx, reads);
    }
    if (rc) { ctx->pool.push_back(b); return rc; }
    b->planes_gen = ctx->hint_in_table ? ctx->hint_gen : 0;      // what this batch's snapshot planes speak of
    b->cand_gen = 0;
    ctx->prepared.push_back(b);
    return FGPU_OK;
}

// Read shards: a FRESHER preview has taken the place of the one the batches were prepared against (fgpu_scan_import_hint again: the table the
// shard below was handed, one hop before this shard's own arrives) -- the snapshot planes of every prepared batch are made again against it,
// off the chain of walks.  When the real table arrives it differs from this preview by the keys ONE shard created, and the walk only has to
// look for those (fgpu_scan_import_table).
int fgpu_scan_refresh_prepared(fgpu_ctx* ctx) {
    if (!ctx) return FGPU_ERR_ARG;
    if (ctx->phase != 2) { ctx->err = "scan_refresh_prepared outside scan_begin/scan_end"; return FGPU_ERR_STATE; }
    if (!ctx->hint_in_table) { ctx->err = "scan_refresh_prepared without a preview in the table (fgpu_scan_import_hint)"; return FGPU_ERR_STATE; }
    FGPU_HIP(hipSetDevice(ctx->prm.device));
    const bool no_sparse = ctx->no_sparse_link;
    for (BatchBufs* b : ctx->prepared) {
        if (b->planes_gen != ctx->hint_gen) {
            if (int rc = fgpu_scan_refresh_planes(ctx, b)) return rc;
            b->planes_gen = ctx->hint_gen;
            b->cand_gen = 0;
        }
        // ... and, while this rank still waits, the plane that lets its walk link a window by visiting candidates only (round 6, k_walk_link_sparse)
        if (!no_sparse && b->cand_gen != ctx->hint_gen)
            if (int rc = fgpu_scan_build_cand(ctx, b)) return rc;
    }
    ctx->launch_stream = ctx->stream;
    FGPU_HIP(fgpu_sync_stream(ctx, ctx->stream));
    return FGPU_OK;
}

int fgpu_scan_walk_prepared(fgpu_ctx* ctx) {
    if (!ctx) return FGPU_ERR_ARG;
    if (ctx->phase != 2) { ctx->err = "scan_walk_prepared outside scan_begin/scan_end"; return FGPU_ERR_STATE; }
    if (ctx->hint_in_table) { ctx->err = "the junction table holds a preview (fgpu_scan_import_hint): import the real table before walking"; return FGPU_ERR_STATE; }
    FGPU_HIP(hipSetDevice(ctx->prm.device));
    int rc = FGPU_OK;
    if (ctx->lazy_failed) return scan_replay(ctx);      // (walks the prepared batches as well)
    // The batches were prepared before their turn: their snapshot planes speak of a table that has since been replaced (the preview by the real
    // table) and walked on, so they are made again by the walk stream right before every batch's walk.  FGPU_PREPARED_REFRESH=overlap
    // (round 5, measured, not the default): batch i + 1's on the main stream WHILE batch i is walked, what batch i creates meanwhile reaching
    // batch i + 1 through the created-key lists as in a streaming scan -- exact (the sharded tests pass with it), and no faster: the 23 ms of
    // plane-making leave the chain, 11 ms of contention and 10 ms of registering three batches' created keys per window enter it (first hop of
    // config 4 on 8 shards 66.2 against 65.4 ms, profiles/r05_hop_refresh_ab.txt).
    static const bool serial_refresh = !(getenv("FGPU_PREPARED_REFRESH") && getenv("FGPU_PREPARED_REFRESH")[0] == 'o');
    if (!serial_refresh && !ctx->prepared.empty()) rc = fgpu_scan_refresh_planes(ctx, ctx->prepared[0]);
    for (size_t i = 0; i < ctx->prepared.size() && !rc; i++) {
        BatchBufs* b = ctx->prepared[i];
        if (i > 0) {   // feedback for the window controller and for the size of the junction table between batches
            if ((rc = pull_counters(ctx))) break;                  // (the walk of batch i - 1 has finished; so have the planes of batch i)
            if (ctx->lazy_failed) { ctx->cur = &ctx->bb_default; return scan_replay(ctx); }
            if (!ctx->prm.walk_window_span) adapt_window(ctx);
            if ((rc = fgpu_scan_reserve(ctx, ctx->counters_host->n_junctions + ctx->scan_imported))) break;
        }
        if (!serial_refresh && i + 1 < ctx->prepared.size() && (rc = fgpu_scan_refresh_planes(ctx, ctx->prepared[i + 1]))) break;
        ctx->cur = b;
        ctx->refresh_snapshot = serial_refresh;
        rc = fgpu_stage_scan_walk(ctx, b->n_pieces);
        ctx->refresh_snapshot = false;
        ctx->walked_pieces += b->n_pieces;
        if (!rc) note_walked(ctx, b);
    }
    ctx->cur = &ctx->bb_default;
    for (BatchBufs* b : ctx->prepared) ctx->pool.push_back(b);
    ctx->prepared.clear();
    return rc;
}

int fgpu_scan_set_eager(fgpu_ctx* ctx, int on) {
    if (!ctx) return FGPU_ERR_ARG;
    if (ctx->phase != 0) { ctx->err = "fgpu_scan_set_eager while a pass is open"; return FGPU_ERR_STATE; }
    ctx->eager_runtime = on != 0;
    return FGPU_OK;
}

// ---- the short pair filter on the device ---------------------------------------------------------------------------------
int fgpu_scan_short_pairs(fgpu_ctx* ctx, uint64_t tai, int32_t n_hash, int32_t lists_to_host) {
    if (!ctx) return FGPU_ERR_ARG;
    if (ctx->phase != 0) { ctx->err = "fgpu_scan_short_pairs while a pass is open"; return FGPU_ERR_STATE; }
    FGPU_HIP(hipSetDevice(ctx->prm.device));
    if (ctx->short_pf) {
        FGPU_HIP(fgpu_sync_stream(ctx, ctx->stream));
        FGPU_HIP(hipFree(ctx->short_pf));
        ctx->short_pf = nullptr;
        ctx->short_pf_tai = 0;
    }
    if (!tai) return FGPU_OK;
    if (!ctx->record_stops) { ctx->err = "fgpu_scan_short_pairs needs FGPU_FLAG_RECORD_STOPS"; return FGPU_ERR_STATE; }
    if (!is_pow2(tai) || tai < 128 || n_hash < 1 || n_hash > 32) { ctx->err = "fgpu_scan_short_pairs: tai must be a power of two >= 128, n_hash 1..32"; return FGPU_ERR_ARG; }
    FGPU_HIP(hipMalloc(&ctx->short_pf, tai / 8));
    FGPU_HIP(hipMemsetAsync(ctx->short_pf, 0, tai / 8, ctx->stream));
    ctx->short_pf_tai = tai;
    ctx->short_pf_hashes = n_hash;
    ctx->short_pf_lists_to_host = lists_to_host != 0;
    return FGPU_OK;
}

int fgpu_scan_short_pairs_download(fgpu_ctx* ctx, uint8_t* out, uint64_t n_bytes) {
    if (!ctx || !out) return FGPU_ERR_ARG;
    if (!ctx->short_pf) { ctx->err = "fgpu_scan_short_pairs_download without fgpu_scan_short_pairs"; return FGPU_ERR_STATE; }
    if (ctx->phase != 0) { ctx->err = "fgpu_scan_short_pairs_download while a pass is open"; return FGPU_ERR_STATE; }
    if (n_bytes != ctx->short_pf_tai / 8) { ctx->err = "fgpu_scan_short_pairs_download: the filter has tai / 8 bytes"; return FGPU_ERR_ARG; }
    FGPU_HIP(hipSetDevice(ctx->prm.device));
    FGPU_HIP(hipMemcpyAsync(out, ctx->short_pf, n_bytes, hipMemcpyDeviceToHost, ctx->stream));
    FGPU_HIP(fgpu_sync_stream(ctx, ctx->stream));
    return FGPU_OK;
}

int fgpu_scan_take_stops(fgpu_ctx* ctx, fgpu_stop* out, uint64_t cap, uint64_t* n_out, int64_t* batch_seq) {
    if (!ctx || !n_out || !batch_seq) return FGPU_ERR_ARG;
    if (!ctx->record_stops) { ctx->err = "fgpu_scan_take_stops needs FGPU_FLAG_RECORD_STOPS"; return FGPU_ERR_STATE; }
    *n_out = 0;
    *batch_seq = -1;
    if (ctx->phase == 2 && ctx->lazy_failed) {
        if (int rc = scan_replay(ctx)) return rc;
    }
    if (ctx->stop_queue.empty() && !ctx->to_harvest.empty()) {
        FGPU_HIP(hipSetDevice(ctx->prm.device));
        int rc = fgpu_scan_harvest(ctx, ctx->to_harvest.front());
        if (rc == FGPU_INTERNAL_REPLAY && ctx->phase == 2) {
            if ((rc = scan_replay(ctx))) return rc;
            if (!ctx->to_harvest.empty()) rc = fgpu_scan_harvest(ctx, ctx->to_harvest.front());
        }
        if (rc) return rc;
    }
    if (ctx->stop_queue.empty()) return FGPU_OK;
    StopBatch& sb = ctx->stop_queue.front();
    *n_out = sb.n;
    *batch_seq = (int64_t)sb.seq;
    if (sb.n > cap || (sb.n && !out)) return FGPU_ERR_CAPACITY;
    if (sb.n) memcpy(out, sb.data, sb.n * sizeof(fgpu_stop));
    ctx->stops_delivered = sb.seq + 1;
    if (sb.data) { StopBatch f; f.data = sb.data; f.cap = sb.cap; ctx->stop_pool.push_back(f); }
    ctx->stop_queue.pop_front();
    return FGPU_OK;
}

int fgpu_scan_end(fgpu_ctx* ctx, fgpu_scan_stats* stats) {
    if (!ctx) return FGPU_ERR_ARG;
    if (ctx->phase != 2) { ctx->err = "scan_end without scan_begin"; return FGPU_ERR_STATE; }
    static const bool dbg_host = getenv("FGPU_DEBUG_HOST") != nullptr;
    static const bool dbg_waits = getenv("FGPU_DEBUG_WAITS") != nullptr;
    struct TellWaits {      // (at the very end of the call: the last walks and harvests are waited for inside it)
        fgpu_ctx* c;
        bool on;
        ~TellWaits() {
            if (!on) return;
            for (const WaitSite& w : c->wait_sites) {
                const char* f = strrchr(w.file, '/');
                fprintf(stderr, "[waits] %-14s:%-5d %6llu times %9.2f ms\n", f ? f + 1 : w.file, w.line, (unsigned long long)w.n, w.ms);
            }
        }
    } tell_waits{ctx, dbg_waits};
    if (dbg_host) {
        fprintf(stderr, "[host] scan: wait for the buffers' last walk %.2f ms, pack issue %.2f, pure stage issue (first half) %.2f, wait for the piece count %.2f, "
                        "pure stage issue (second half) %.2f, walk issue %.2f\n", ctx->host_ms[0], ctx->host_ms[1], ctx->host_ms[2], ctx->host_ms[3], ctx->host_ms[4], ctx->host_ms[5]);
        for (double& v : ctx->host_ms) v = 0;
    }
    int rc = pull_counters(ctx);
    if (!rc && ctx->lazy_failed) {                          // the last walks met a preview they could not repair: scan again before closing
        rc = scan_replay(ctx);
        if (!rc) rc = pull_counters(ctx);
    }
    while (!rc && !ctx->to_harvest.empty()) rc = fgpu_scan_harvest(ctx, ctx->to_harvest.front());   // every walk has finished
    if (!rc) rc = fgpu_long_pairs_close(ctx);              // the last batch of lists: its rounds' outcome, a first end left without a mate
    ctx->phase = 0;
    ctx->journal_on = false;
    if (!rc && !ctx->prm.walk_window_span && ctx->scan_windows > 8) ctx->settled_span = ctx->window_span;
    if (rc) return rc;
    const DevCounters& c = *ctx->counters_host;
    fgpu_scan_stats& s = ctx->scan_stats;
    s.unambiguous_reads = c.segments + ctx->carried.unambiguous_reads;
    s.reads_no_errors = c.pieces + ctx->carried.reads_no_errors;
    s.nb_jcheck_kmer = c.nb_jcheck + ctx->carried.nb_jcheck_kmer;
    s.nb_no_juncs = c.nb_no_juncs + ctx->carried.nb_no_juncs;
    s.nb_processed = c.nb_processed + ctx->carried.nb_processed;
    s.nb_skipped = c.nb_skipped + ctx->carried.nb_skipped;
    s.n_junctions = c.n_junctions + ctx->scan_imported;
    s.kmers = c.kmers + ctx->carried.kmers;
    s.reads_processed += ctx->carried.reads_processed;
    s.walk_windows = ctx->scan_windows;
    s.walk_followers = c.followers;
    s.walk_max_cluster = c.max_cluster;
    s.flag_positions = c.flag_positions;
    s.piece_positions = c.piece_positions;
    s.valid_reused = c.valid_reused;
    s.flags_filled = c.flags_filled;
    s.walk_parallel = c.walk_parallel;
    memset(&ctx->carried, 0, sizeof(ctx->carried));
    if (stats) *stats = s;
    return FGPU_OK;
}

int fgpu_scan_junction_count(fgpu_ctx* ctx, uint64_t* n) {
    if (!ctx || !n) return FGPU_ERR_ARG;
    if (int rc = sync_all(ctx)) return rc;
    return fgpu_scan_download_impl(ctx, nullptr, nullptr, 0, n);
}

int fgpu_scan_download_junctions(fgpu_ctx* ctx, uint64_t* keys, fgpu_junction* recs, uint64_t cap, uint64_t* n_out) {
    if (!ctx || !keys || !recs || !n_out) return FGPU_ERR_ARG;
    if (!ctx->jkeys) { *n_out = 0; return FGPU_OK; }
    if (int rc = sync_all(ctx)) return rc;
    return fgpu_scan_download_impl(ctx, keys, recs, cap, n_out);
}

int fgpu_scan_table_entries(fgpu_ctx* ctx, uint64_t* n_entries) { return fgpu_scan_junction_count(ctx, n_entries); }

int fgpu_scan_dump_order(fgpu_ctx* ctx, const uint64_t* rehash_counts, const uint64_t* rehash_buckets, uint64_t n_rehashes, uint64_t n, uint32_t* order) {
    if (!ctx || !rehash_counts || !rehash_buckets || !n_rehashes || (n && !order)) return FGPU_ERR_ARG;
    if (ctx->phase != 0) { ctx->err = "fgpu_scan_dump_order while a pass is open"; return FGPU_ERR_STATE; }
    if (!ctx->dl_keys_n || n > ctx->dl_keys_n) { ctx->err = "fgpu_scan_dump_order works on the keys of the last fgpu_scan_download_junctions (at most that many)"; return FGPU_ERR_STATE; }
    FGPU_HIP(hipSetDevice(ctx->prm.device));
    return fgpu_scan_dump_order_impl(ctx, (const uint64_t*)ctx->dl_keys.p, rehash_counts, rehash_buckets, n_rehashes, n, order);
}

int fgpu_scan_export_table(fgpu_ctx* ctx, void* dev_buf, uint64_t buf_bytes, uint64_t* n_entries) {
    if (!ctx || !dev_buf || !n_entries) return FGPU_ERR_ARG;
    uint64_t n = 0;
    int rc = sync_all(ctx);   // walks still running on the walk stream are part of the table
    if (!rc) rc = fgpu_scan_junction_count(ctx, &n);
    if (rc) return rc;
    if (buf_bytes < n * FGPU_TABLE_ENTRY_BYTES) { ctx->err = "export buffer too small"; return FGPU_ERR_CAPACITY; }
    if ((rc = fgpu_ensure(ctx, &ctx->export_stamps, n * 8 + 8))) return rc;
    return fgpu_scan_export_impl(ctx, dev_buf, n, (uint64_t*)ctx->export_stamps.p, n_entries);
}

int fgpu_scan_import_table(fgpu_ctx* ctx, const void* dev_buf, uint64_t n_entries, const fgpu_scan_stats* carried) {
    if (!ctx || (n_entries && !dev_buf)) return FGPU_ERR_ARG;
    if (ctx->phase != 2) { ctx->err = "import_table outside scan_begin/scan_end"; return FGPU_ERR_STATE; }
    int rc = sync_all(ctx);
    if (rc) return rc;
    ctx->delta_ready = false;
    if (ctx->hint_in_table) {   // the preview has done its work (the prepared batches' planes): the real table takes its place
        if ((rc = fgpu_scan_clear_table(ctx))) return rc;
        ctx->hint_in_table = false;
        // If every prepared batch's planes speak of the newest preview and this table is a LATER STATE of it (the caller's promise, checked by
        // counting: the entries with a creation stamp beyond the preview's are exactly the surplus), the walk merges the new keys into the planes
        // through a filter of just those keys instead of probing the whole table's filter at every position again (22 -> ~7 ms per 25 M reads).
        static const bool no_delta = getenv("FGPU_NO_DELTA_REFRESH") != nullptr;
        bool all = ctx->hint_gen != 0 && !ctx->prepared.empty() && !no_delta && n_entries >= ctx->hint_n;
        for (BatchBufs* b : ctx->prepared) all = all && b->planes_gen == ctx->hint_gen;
        const uint64_t surplus = all ? n_entries - ctx->hint_n : 0;
        if (all && surplus <= (1ULL << 23)) {
            uint64_t bits = 1ULL << 20;      // (room for what this shard's own batches will add: they join the filter as they are walked)
            // (32 bits of filter per new key, swept in round 6 on config 4's hop, FGPU_DELTA_FILTER_MULT = 8 / 16 / 32 / 64: the merge 16.4 / 14.4 / 13.0 / 13.9 ms, the
            // link by the candidate plane -- which takes the filter's hits -- 11.6 / 9.7 / 8.7 / 8.2 ms, the hop 71.1 / 67.6 / 64.8 / 66.0 ms: false hits cost more than cache misses)
            static const uint64_t mult = getenv("FGPU_DELTA_FILTER_MULT") ? (uint64_t)std::max(2, atoi(getenv("FGPU_DELTA_FILTER_MULT"))) : 32;
            while (bits < mult * surplus) bits <<= 1;
            if ((rc = fgpu_ensure(ctx, &ctx->delta_filter, bits / 8))) return rc;
            FGPU_HIP(hipMemsetAsync(ctx->delta_filter.p, 0, bits / 8, ctx->stream));
            uint64_t max_seq = 0, newer = 0, digest[2] = {0, 0};
            if ((rc = fgpu_scan_import_probe(ctx, dev_buf, n_entries, ctx->hint_max_seq, (uint32_t*)ctx->delta_filter.p, bits, &max_seq, &newer, digest))) return rc;
            // a later state of the preview: as many newer entries as the surplus, AND the others are the preview's keys (their digest)
            if (newer == surplus && digest[0] == ctx->hint_digest[0] && digest[1] == ctx->hint_digest[1]) {
                ctx->delta_ready = true;
                ctx->delta_filter_bits = bits;
                ctx->delta_keys = newer;
            }
        }
    }
    if ((rc = fgpu_scan_reserve(ctx, ctx->counters_host->n_junctions + ctx->scan_imported + n_entries))) return rc;
    rc = fgpu_scan_import_impl(ctx, dev_buf, n_entries);
    if (rc) return rc;
    if (ctx->journal_on) {   // a replay starts from this table again
        if ((rc = fgpu_ensure(ctx, &ctx->import_copy, n_entries * FGPU_TABLE_ENTRY_BYTES + 16))) return rc;
        if (n_entries) FGPU_HIP(hipMemcpyAsync(ctx->import_copy.p, dev_buf, n_entries * FGPU_TABLE_ENTRY_BYTES, hipMemcpyDeviceToDevice, ctx->stream));
        ctx->import_n = n_entries;
        ctx->have_import = true;
        ctx->import_has_carried = carried != nullptr;
        if (carried) ctx->import_carried = *carried;
    }
    // the import runs on the main stream, the ordered walk on the walk stream behind events recorded BEFORE this call
    // (end of each prepared batch's pure stage): without this wait the walk would start on a half-imported table
    FGPU_HIP(fgpu_sync_stream(ctx, ctx->stream));
    ctx->scan_imported += n_entries;
    if (carried) ctx->carried = *carried;
    // creation stamps of this shard must sort after everything imported
    ctx->scan_piece_base = std::max<uint64_t>(ctx->scan_piece_base, carried ? carried->reads_no_errors : 0);
    return FGPU_OK;
}

int fgpu_scan_import_hint(fgpu_ctx* ctx, const void* dev_buf, uint64_t n_entries) {
    if (!ctx || (n_entries && !dev_buf)) return FGPU_ERR_ARG;
    if (ctx->phase != 2) { ctx->err = "import_hint outside scan_begin/scan_end"; return FGPU_ERR_STATE; }
    if (ctx->scan_windows || ctx->scan_imported) {
        ctx->err = "import_hint comes before any walk and before the real table (batches prepared earlier have simply seen an empty table)";
        return FGPU_ERR_STATE;
    }
    int rc = sync_all(ctx);
    if (rc) return rc;
    if (ctx->hint_in_table && (rc = fgpu_scan_clear_table(ctx))) return rc;      // a fresher preview takes the place of an older one
    if ((rc = fgpu_scan_reserve(ctx, n_entries))) return rc;
    if ((rc = fgpu_scan_import_impl(ctx, dev_buf, n_entries))) return rc;
    uint64_t newer = 0;
    ctx->hint_max_seq = 0;
    if ((rc = fgpu_scan_import_probe(ctx, dev_buf, n_entries, ~0ULL, nullptr, 0, &ctx->hint_max_seq, &newer, ctx->hint_digest))) return rc;
    FGPU_HIP(fgpu_sync_stream(ctx, ctx->stream));
    ctx->hint_in_table = true;
    ctx->hint_n = n_entries;
    ctx->hint_gen++;
    return FGPU_OK;
}

// ---- probes ----------------------------------------------------------------------------------------------------
int fgpu_probe_hash(fgpu_ctx* ctx, const uint64_t* kmers_host, uint64_t n, uint64_t* canon_out, uint64_t* hA_out, uint64_t* hB_out) {
    if (!ctx || !kmers_host || !canon_out || !hA_out || !hB_out) return FGPU_ERR_ARG;
    if (!n) return FGPU_OK;
    DevBuf& b = ctx->probe_buf;
    int rc = fgpu_ensure(ctx, &b, n * 32);
    if (rc) return rc;
    uint64_t* d = (uint64_t*)b.p;
    FGPU_HIP(hipMemcpyAsync(d, kmers_host, n * 8, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = fgpu_util_probe_hash(ctx, d, n, d + n, d + 2 * n, d + 3 * n))) return rc;
    FGPU_HIP(hipMemcpyAsync(canon_out, d + n, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    FGPU_HIP(hipMemcpyAsync(hA_out, d + 2 * n, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    FGPU_HIP(hipMemcpyAsync(hB_out, d + 3 * n, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    FGPU_HIP(fgpu_sync_stream(ctx, ctx->stream));
    return FGPU_OK;
}

int fgpu_probe_contains(fgpu_ctx* ctx, int which, const uint64_t* canon_host, uint64_t n, uint8_t* out) {
    if (!ctx || !canon_host || !out || !bloom_ptr(ctx, which)) return FGPU_ERR_ARG;
    if (!n) return FGPU_OK;
    DevBuf& b = ctx->probe_buf;
    int rc = fgpu_ensure(ctx, &b, n * 16);
    if (rc) return rc;
    uint64_t* d = (uint64_t*)b.p;
    FGPU_HIP(hipMemcpyAsync(d, canon_host, n * 8, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = fgpu_util_probe_contains(ctx, bloom_ptr(ctx, which), d, n, (unsigned char*)(d + n)))) return rc;
    FGPU_HIP(hipMemcpyAsync(out, d + n, n, hipMemcpyDeviceToHost, ctx->stream));
    FGPU_HIP(fgpu_sync_stream(ctx, ctx->stream));
    return FGPU_OK;
}

// ---- profiling ---------------------------------------------------------------------------------------------------
static int probe_stage3(fgpu_ctx* ctx, const uint64_t* kmers_host, uint64_t n, int mode, int8_t* out) {
    if (!ctx || !kmers_host || !out) return FGPU_ERR_ARG;
    if (!n) return FGPU_OK;
    DevBuf& b = ctx->probe_buf;
    int rc = fgpu_ensure(ctx, &b, n * 16);
    if (rc) return rc;
    uint64_t* d = (uint64_t*)b.p;
    FGPU_HIP(hipMemcpyAsync(d, kmers_host, n * 8, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = fgpu_util_probe_stage3(ctx, d, n, mode, (signed char*)(d + n)))) return rc;
    FGPU_HIP(hipMemcpyAsync(out, d + n, n, hipMemcpyDeviceToHost, ctx->stream));
    FGPU_HIP(fgpu_sync_stream(ctx, ctx->stream));
    return FGPU_OK;
}
int fgpu_probe_jcheck(fgpu_ctx* ctx, const uint64_t* kmers_host, uint64_t n, int8_t* out) { return probe_stage3(ctx, kmers_host, n, 0, out); }
int fgpu_probe_valid_extension(fgpu_ctx* ctx, const uint64_t* kmers_host, uint64_t n, int8_t* out) { return probe_stage3(ctx, kmers_host, n, 1, out); }
int fgpu_probe_bloom_junction(fgpu_ctx* ctx, const uint64_t* kmers_host, uint64_t n, int8_t* out) { return probe_stage3(ctx, kmers_host, n, 2, out); }

int fgpu_diag_host_waits(fgpu_ctx* ctx, uint64_t* waits, double* ms) {
    if (!ctx || !waits) return FGPU_ERR_ARG;
    *waits = ctx->host_waits;
    if (ms) *ms = ctx->host_wait_ms;
    return FGPU_OK;
}

int fgpu_diag_scan_replays(fgpu_ctx* ctx, uint64_t* replays) {
    if (!ctx || !replays) return FGPU_ERR_ARG;
    *replays = ctx->scan_replays;
    return FGPU_OK;
}

int fgpu_diag_late_flags(fgpu_ctx* ctx, uint64_t out[3]) {   // after fgpu_scan_end, see the header
    if (!ctx || !out) return FGPU_ERR_ARG;
    for (int i = 0; i < 3; i++) out[i] = ctx->late_acc[i] + ctx->counters_host->late_n[i];
    return FGPU_OK;
}

int fgpu_diag_walk_probe(fgpu_ctx* ctx, uint64_t out[4]) {   // after fgpu_scan_end: probed pieces by outcome (k_walk_par)
    if (!ctx || !out) return FGPU_ERR_ARG;
    for (int i = 0; i < 4; i++) out[i] = ctx->counters_host->par_probe[i];
    if (getenv("FGPU_KO_TIMING_PRINT")) {
        fprintf(stderr, "[fgpu] k_walk_ko ticks (10 ns):");
        for (int i = 0; i < 8; i++) fprintf(stderr, " %llu", ctx->counters_host->ko_time[i]);
        fprintf(stderr, "\n");
    }
    return FGPU_OK;
}

// measurement builds (-DFGPU_KO_TRACE): one record per piece the key-ordered walk has walked since the context was made -- {global piece
// number, start, end (10 ns ticks of the device's constant clock), ticks waited for turns | lk positions << 48}; *n = 0 in ordinary builds
int fgpu_diag_ko_trace(fgpu_ctx* ctx, uint64_t* out, uint64_t cap_records, uint64_t* n) {
    if (!ctx || !n) return FGPU_ERR_ARG;
    *n = 0;
    if (!ctx->ko_trace.p) return FGPU_OK;
    if (int rc = sync_all(ctx)) return rc;
    unsigned long long cnt = 0;
    FGPU_HIP(hipMemcpy(&cnt, ctx->ko_trace.p, 8, hipMemcpyDeviceToHost));
    if (cnt > (1ULL << 21)) cnt = 1ULL << 21;
    if (cnt > cap_records) cnt = cap_records;
    if (cnt && out) FGPU_HIP(hipMemcpy(out, (const char*)ctx->ko_trace.p + 32, cnt * 32, hipMemcpyDeviceToHost));
    *n = cnt;
    return FGPU_OK;
}

// the per-step time stamps of the stamped pieces (one in 16): 4096 x 1024 words, see KO_STAMP in scan_walk.hip
int fgpu_diag_ko_stamps(fgpu_ctx* ctx, uint64_t* out, uint64_t cap_words, uint64_t* n_pieces) {
    if (!ctx || !n_pieces) return FGPU_ERR_ARG;
    *n_pieces = 0;
    if (!ctx->ko_trace.p || ctx->ko_trace.bytes < ((1ULL << 23) + 4096 * 1024) * 8) return FGPU_OK;
    if (int rc = sync_all(ctx)) return rc;
    unsigned long long cnt = 0;
    FGPU_HIP(hipMemcpy(&cnt, (const char*)ctx->ko_trace.p + 8, 8, hipMemcpyDeviceToHost));
    if (cnt > 4096) cnt = 4096;
    if (cnt * 1024 > cap_words) cnt = cap_words / 1024;
    if (cnt && out) FGPU_HIP(hipMemcpy(out, (const char*)ctx->ko_trace.p + (8ULL << 23), cnt * 1024 * 8, hipMemcpyDeviceToHost));
    *n_pieces = cnt;
    return FGPU_OK;
}

int fgpu_diag_load_split(fgpu_ctx* ctx, uint64_t* in_mark, uint64_t* pending) {
    if (!ctx || !in_mark || !pending) return FGPU_ERR_ARG;
    *in_mark = ctx->load_mark_hits;
    *pending = ctx->load_mark_pending;
    return FGPU_OK;
}

int fgpu_diag_ovw(fgpu_ctx* ctx, uint64_t out[6]) {
    if (!ctx || !out) return FGPU_ERR_ARG;
    for (int i = 0; i < 4; i++) out[i] = ctx->counters_host->ovw[i];   // as of the scan's last synchronising call (fgpu_scan_end)
    out[4] = ctx->counters_host->ovw_kept;
    out[5] = ctx->counters_host->ko_overflows;
    return FGPU_OK;
}

int fgpu_diag_prepared_refresh(fgpu_ctx* ctx, uint64_t out[4]) {
    if (!ctx || !out) return FGPU_ERR_ARG;
    out[0] = ctx->refresh_full;
    out[1] = ctx->refresh_delta;
    out[2] = ctx->delta_keys;
    out[3] = ctx->refresh_mismatch;
    return FGPU_OK;
}

int fgpu_diag_sparse_link(fgpu_ctx* ctx, uint64_t out[2]) {
    if (!ctx || !out) return FGPU_ERR_ARG;
    out[0] = ctx->sparse_links;
    out[1] = ctx->full_links;
    return FGPU_OK;
}

int fgpu_diag_ovw_tables(fgpu_ctx* ctx, uint64_t* high_water, uint64_t* capacity) {
    if (!ctx || !high_water || !capacity) return FGPU_ERR_ARG;
    *high_water = ctx->counters_host->ovw_fill;          // as of the scan's last synchronising call
    *capacity = 1ULL << ctx->ovw_ev_log2;
    return FGPU_OK;
}

int fgpu_kernel_times(fgpu_ctx* ctx, fgpu_kernel_time* out, int cap) {
    if (!ctx) return 0;
    if (fgpu_prof_collect(ctx) != FGPU_OK) return 0;
    int n = (int)ctx->kstats.size();
    for (int i = 0; i < n && i < cap && out; i++) {
        memset(&out[i], 0, sizeof(out[i]));
        strncpy(out[i].name, ctx->kstats[i].name.c_str(), sizeof(out[i].name) - 1);
        out[i].launches = ctx->kstats[i].launches;
        out[i].total_ms = ctx->kstats[i].total_ms;
    }
    return n;
}

int fgpu_profile_enable(fgpu_ctx* ctx, int on) {
    if (!ctx) return FGPU_ERR_ARG;
    if (ctx->phase != 0) { ctx->err = "fgpu_profile_enable while a pass is open"; return FGPU_ERR_STATE; }
    int rc = fgpu_prof_collect(ctx);
    ctx->profile = on != 0;
    return rc;
}

int fgpu_kernel_times_reset(fgpu_ctx* ctx) {
    if (!ctx) return FGPU_ERR_ARG;
    int rc = fgpu_prof_collect(ctx);
    ctx->kstats.clear();
    return rc;
}

}  // extern "C"
