// faucet_gpu_stub.cpp — TEST INFRASTRUCTURE ONLY: the C ABI of include/faucet_gpu.h answered on the CPU by the parity oracle.
//
// NOT part of the product and never shipped: libfaucet_gpu.so has no CPU path (fgpu_create fails without a gfx950 device).  This file
// exists so that the HOST code around the ABI can be exercised where there is no GPU and under sanitizers (GPU AddressSanitizer is not
// available on the pool):
//   * faucet_amd/host/faucet_main.cpp (reader threads, read-ahead, pair-filter worker, formatter threads) built with
//     -fsanitize=thread and -fsanitize=address,undefined against this stub (tests/test_host_sanitizers.py);
//   * integration/faucet_binding.cpp LINKED into the compiled reference (oracle/Makefile, target ref_stub) and run on the CPU, its
//     contig-graph stage included (tests/test_binding_link.py).
// Only what those two callers use is implemented; every other entry point reports FGPU_ERR_STATE.  Only tests/ builds or links it.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/faucet_gpu.h"
#include "../../oracle/faucet_oracle.h"

struct StubStops {
    int64_t seq;
    std::vector<fgpu_stop> stops;
};

struct fgpu_ctx {
    fgpu_params prm;
    fo_bloom *b1, *b2, *short_pf, *long_pf;
    int long_mode;                           // fgpu_scan_long_pairs: FGPU_LONG_PAIRS_*
    bool first_end;                          // scanReads' firstEnd toggle (src/ReadScanner.cpp:303,350)
    std::vector<uint64_t> back1;             // the first end's list while its mate is awaited
    uint64_t empty_count, not_empty_count;
    fo_scanner* sc;
    uint64_t short_tai;
    int short_lists_to_host;
    std::string err;
    fgpu_load_stats lst;
    uint64_t scan_reads, scan_kmers;
    int phase;                               // 0 idle, 1 loading, 2 scanning
    int64_t batch_seq;
    std::deque<StubStops> queue;
    // fgpu_text_split: the batch handed out points into these
    std::vector<char> text;
    std::vector<uint64_t> starts, offsets;
    std::vector<char> flat;                  // a batch with `starts` made contiguous for the oracle
    std::vector<uint64_t> flat_offs;
    struct Prepared { std::vector<char> bases; std::vector<uint64_t> offs; uint64_t n; };
    std::deque<Prepared> prepared;           // fgpu_scan_prepare: batches waiting for fgpu_scan_walk_prepared
    uint64_t carried_reads;                  // reads_processed handed over by the shard below (fgpu_scan_import_table)
};

static thread_local std::string g_create_error;

static int fail(fgpu_ctx* c, int rc, const char* what) {
    if (c) c->err = what; else g_create_error = what;
    return rc;
}

extern "C" {

int fgpu_abi_version(void) { return FGPU_ABI_VERSION; }
int fgpu_device_count(void) { return 1; }   // (the stand-in IS the device)

int fgpu_create(const fgpu_params* p, fgpu_ctx** out) {
    if (!p || !out) return FGPU_ERR_ARG;
    if (p->k < 1 || p->k > 31 || p->n_hash < 1 || p->n_hash > 10 || p->tai < 128 || (p->tai & (p->tai - 1))) return fail(NULL, FGPU_ERR_ARG, "bad parameters");
    fgpu_ctx* c = new fgpu_ctx();
    c->prm = *p;
    c->b1 = fo_bloom_new(p->tai, p->n_hash);
    c->b2 = fo_bloom_new(p->tai, p->n_hash);
    c->short_pf = NULL;
    c->long_pf = NULL;
    c->long_mode = FGPU_LONG_PAIRS_OFF;
    c->first_end = true;
    c->empty_count = c->not_empty_count = 0;
    c->sc = NULL;
    c->short_tai = 0;
    c->short_lists_to_host = 1;
    memset(&c->lst, 0, sizeof(c->lst));
    c->scan_reads = c->scan_kmers = 0;
    c->phase = 0;
    c->batch_seq = 0;
    c->carried_reads = 0;
    *out = c;
    return FGPU_OK;
}

void fgpu_destroy(fgpu_ctx* c) {
    if (!c) return;
    if (c->sc) fo_scanner_free(c->sc);
    if (c->short_pf) fo_bloom_free(c->short_pf);
    if (c->long_pf) fo_bloom_free(c->long_pf);
    fo_bloom_free(c->b1);
    fo_bloom_free(c->b2);
    delete c;
}

const char* fgpu_last_error(const fgpu_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }
int fgpu_synchronize(fgpu_ctx* c) { return c ? FGPU_OK : FGPU_ERR_ARG; }

void* fgpu_host_alloc(uint64_t bytes) { return malloc(bytes ? bytes : 1); }
void fgpu_host_free(void* p) { free(p); }

// the reference's reading loop over a chunk of text (utils/Bloom.cpp:280-282,340; src/ReadScanner.cpp:306-308,349)
int fgpu_text_reserve(fgpu_ctx* c, uint64_t) { return c ? FGPU_OK : FGPU_ERR_ARG; }
int fgpu_text_split(fgpu_ctx* c, const char* text, uint64_t nbytes, int text_on_device, int fastq, int final_chunk, fgpu_reads* out,
                    uint64_t* consumed) {
    if (!c || !out || !consumed || text_on_device) return FGPU_ERR_ARG;
    c->text.assign(text, text + nbytes);
    c->starts.clear();
    c->offsets.assign(1, 0);
    const uint64_t lines_per_record = fastq ? 4 : 2;
    uint64_t pos = 0, used = 0;
    for (;;) {
        // the record's lines: [begin, end) without the newline; a line that ends with the text (no newline) only counts in the final chunk
        uint64_t p = pos, seq_b = 0, seq_e = 0, have = 0;
        bool complete = true;
        for (uint64_t l = 0; l < lines_per_record; l++) {
            if (p >= nbytes) {                      // getline fails: the header's failure ends the loop, later ones leave the line empty
                if (l == 0) { complete = false; break; }
                if (!final_chunk) { complete = false; break; }
                if (l == 1) seq_b = seq_e = nbytes;
                have++;
                continue;
            }
            const char* nl = (const char*)memchr(c->text.data() + p, '\n', nbytes - p);
            if (!nl && !final_chunk) { complete = false; break; }
            const uint64_t e = nl ? (uint64_t)(nl - c->text.data()) : nbytes;
            if (l == 1) { seq_b = p; seq_e = e; }
            p = nl ? e + 1 : nbytes;
            have++;
        }
        if (!complete || have == 0) break;
        c->starts.push_back(seq_b);
        c->offsets.push_back(c->offsets.back() + (seq_e - seq_b));
        pos = used = p;
        if (pos >= nbytes) break;
    }
    *consumed = final_chunk ? nbytes : used;
    memset(out, 0, sizeof(*out));
    out->bases = c->text.data();
    out->offsets = c->offsets.data();
    out->starts = c->starts.empty() ? NULL : c->starts.data();
    out->n_reads = c->starts.size();
    out->on_device = 1;                         // what the product returns; the stub's "device" is host memory
    return FGPU_OK;
}

// a batch as the contiguous (bases, offsets) pair the oracle takes
static void flatten(fgpu_ctx* c, const fgpu_reads* r, const char** bases, const uint64_t** offs) {
    if (!r->starts) { *bases = r->bases; *offs = r->offsets; return; }
    c->flat.clear();
    c->flat_offs.assign(1, 0);
    for (uint64_t i = 0; i < r->n_reads; i++) {
        const uint64_t len = r->offsets[i + 1] - r->offsets[i];
        c->flat.insert(c->flat.end(), r->bases + r->starts[i], r->bases + r->starts[i] + len);
        c->flat_offs.push_back(c->flat.size());
    }
    if (c->flat.empty()) c->flat.push_back('N');
    *bases = c->flat.data();
    *offs = c->flat_offs.data();
}

int fgpu_load_begin(fgpu_ctx* c, int flags) {
    if (!c) return FGPU_ERR_ARG;
    if (!(flags & FGPU_LOAD_KEEP_CARRY)) memset(fo_bloom_bits(c->b1), 0, fo_bloom_nbytes(c->b1));
    memset(fo_bloom_bits(c->b2), 0, fo_bloom_nbytes(c->b2));
    memset(&c->lst, 0, sizeof(c->lst));
    c->phase = 1;
    return FGPU_OK;
}

int fgpu_load_batch(fgpu_ctx* c, const fgpu_reads* r) {
    if (!c || !r) return FGPU_ERR_ARG;
    if (c->phase != 1) return fail(c, FGPU_ERR_STATE, "load_batch outside load_begin/load_end");
    const char* bases;
    const uint64_t* offs;
    flatten(c, r, &bases, &offs);
    fo_load_stats st;
    if (c->prm.flags & FGPU_FLAG_MERCY) fo_load_two_filters_mercy(c->b1, c->b2, bases, offs, r->n_reads, c->prm.k, &st);
    else fo_load_two_filters(c->b1, c->b2, bases, offs, r->n_reads, c->prm.k, &st);
    c->lst.reads_processed += st.reads_processed;
    c->lst.unambiguous_reads += st.unambiguous_reads;
    c->lst.kmers += st.kmers;
    c->lst.to_bloo2 += st.to_bloo2;
    return FGPU_OK;
}

int fgpu_load_end(fgpu_ctx* c, fgpu_load_stats* st) {
    if (!c) return FGPU_ERR_ARG;
    if (c->phase != 1) return fail(c, FGPU_ERR_STATE, "load_end without load_begin");
    if (st) *st = c->lst;
    c->phase = 0;
    return FGPU_OK;
}

static fo_bloom* which_bloom(fgpu_ctx* c, int which) { return which == FGPU_BLOO1 ? c->b1 : which == FGPU_BLOO2 ? c->b2 : NULL; }

int fgpu_bloom_download(fgpu_ctx* c, int which, uint8_t* out, uint64_t nbytes) {
    fo_bloom* b = c ? which_bloom(c, which) : NULL;
    if (!b || !out || nbytes != fo_bloom_nbytes(b)) return FGPU_ERR_ARG;
    memcpy(out, fo_bloom_bits(b), nbytes);
    return FGPU_OK;
}
int fgpu_bloom_download_begin(fgpu_ctx* c, int which, uint8_t* out, uint64_t nbytes) { return fgpu_bloom_download(c, which, out, nbytes); }
int fgpu_bloom_download_wait(fgpu_ctx* c) { return c ? FGPU_OK : FGPU_ERR_ARG; }
int fgpu_bloom_upload(fgpu_ctx* c, int which, const uint8_t* in, uint64_t nbytes) {
    fo_bloom* b = c ? which_bloom(c, which) : NULL;
    if (!b || !in || nbytes != fo_bloom_nbytes(b)) return FGPU_ERR_ARG;
    memcpy(fo_bloom_bits(b), in, nbytes);
    return FGPU_OK;
}
int fgpu_bloom_weight(fgpu_ctx* c, int which, float* w) {
    fo_bloom* b = c ? which_bloom(c, which) : NULL;
    if (!b || !w) return FGPU_ERR_ARG;
    *w = fo_bloom_weight(b);
    return FGPU_OK;
}

int fgpu_scan_short_pairs(fgpu_ctx* c, uint64_t tai, int32_t n_hash, int32_t lists_to_host) {
    if (!c) return FGPU_ERR_ARG;
    if (c->phase) return fail(c, FGPU_ERR_STATE, "only between passes");
    if (tai && !(c->prm.flags & FGPU_FLAG_RECORD_STOPS)) return fail(c, FGPU_ERR_STATE, "fgpu_scan_short_pairs needs FGPU_FLAG_RECORD_STOPS");
    if (c->short_pf) { fo_bloom_free(c->short_pf); c->short_pf = NULL; }
    c->short_tai = tai;
    c->short_lists_to_host = lists_to_host;
    if (tai) c->short_pf = fo_bloom_new(tai, n_hash);
    return FGPU_OK;
}
int fgpu_scan_short_pairs_download(fgpu_ctx* c, uint8_t* out, uint64_t n) {
    if (!c || !out || !c->short_pf || n != fo_bloom_nbytes(c->short_pf)) return FGPU_ERR_ARG;
    memcpy(out, fo_bloom_bits(c->short_pf), n);
    return FGPU_OK;
}

// scanReads' paired-end loop (src/ReadScanner.cpp:317-343) as the product applies it on the device; here, the loop itself on the oracle's Bloom
int fgpu_scan_long_pairs(fgpu_ctx* c, uint64_t tai, int32_t n_hash, int32_t mode) {
    if (!c) return FGPU_ERR_ARG;
    if (c->phase) return fail(c, FGPU_ERR_STATE, "only between passes");
    if (mode != FGPU_LONG_PAIRS_OFF && !(c->prm.flags & FGPU_FLAG_RECORD_STOPS)) return fail(c, FGPU_ERR_STATE, "fgpu_scan_long_pairs needs FGPU_FLAG_RECORD_STOPS");
    if (c->long_pf) { fo_bloom_free(c->long_pf); c->long_pf = NULL; }
    c->long_mode = mode;
    if (mode == FGPU_LONG_PAIRS_FILTER) {
        if (!tai || (tai & (tai - 1)) || n_hash < 1) return fail(c, FGPU_ERR_ARG, "bad long pair filter shape");
        if (getenv("FGPU_DEBUG_LONG_PAIRS_NOMEM")) {      // (the product's knob: "the filter's working state does not fit the device" -- the hosts' own loop takes over)
            c->long_mode = FGPU_LONG_PAIRS_OFF;
            return fail(c, FGPU_ERR_NOMEM, "stub: fgpu_scan_long_pairs told to fail (FGPU_DEBUG_LONG_PAIRS_NOMEM)");
        }
        c->long_pf = fo_bloom_new(tai, n_hash);
    }
    return FGPU_OK;
}
int fgpu_scan_long_pairs_download(fgpu_ctx* c, uint8_t* out, uint64_t n, uint64_t* empty_count, uint64_t* not_empty_count) {
    if (!c || c->long_mode == FGPU_LONG_PAIRS_OFF) return FGPU_ERR_ARG;
    if (out) {
        if (!c->long_pf || n != fo_bloom_nbytes(c->long_pf)) return FGPU_ERR_ARG;
        memcpy(out, fo_bloom_bits(c->long_pf), n);
    }
    if (empty_count) *empty_count = c->empty_count;
    if (not_empty_count) *not_empty_count = c->not_empty_count;
    return FGPU_OK;
}
int fgpu_diag_long_pairs(fgpu_ctx* c, uint64_t out[6]) {
    if (!c || !out) return FGPU_ERR_ARG;
    memset(out, 0, 6 * sizeof(uint64_t));
    return FGPU_OK;
}
static void stub_paired_end(fgpu_ctx* c, const uint64_t* ext, uint64_t n) {
    if (c->long_mode == FGPU_LONG_PAIRS_OFF) return;
    if (c->first_end) {
        c->back1.assign(ext, ext + n);
    } else if (!c->back1.empty() && n) {
        c->not_empty_count++;
        for (size_t a = 0; c->long_pf && a < c->back1.size(); a++) {
            bool paired = false;
            for (uint64_t b = 0; b < n && !paired; b++) paired = fo_bloom_contains_pair(c->long_pf, c->back1[a], ext[b], c->prm.k) != 0;
            if (!paired) fo_bloom_add_pair(c->long_pf, c->back1[a], ext[0], c->prm.k);
        }
    } else {
        c->empty_count++;
    }
    c->first_end = !c->first_end;
}

int fgpu_scan_begin(fgpu_ctx* c) {
    if (!c) return FGPU_ERR_ARG;
    if (c->sc) fo_scanner_free(c->sc);
    if (c->short_pf) memset(fo_bloom_bits(c->short_pf), 0, fo_bloom_nbytes(c->short_pf));
    if (c->long_pf) memset(fo_bloom_bits(c->long_pf), 0, fo_bloom_nbytes(c->long_pf));
    c->first_end = true;
    c->back1.clear();
    c->empty_count = c->not_empty_count = 0;
    c->sc = fo_scanner_new(c->prm.k, c->prm.j, c->prm.max_spacer_dist, c->b2, c->short_pf, NULL);
    c->scan_reads = c->scan_kmers = 0;
    c->batch_seq = 0;
    c->queue.clear();
    c->prepared.clear();
    c->carried_reads = 0;
    c->phase = 2;
    return FGPU_OK;
}

static int stub_scan_flat(fgpu_ctx* c, const char* bases, const uint64_t* offs, uint64_t n_reads) {
    const bool record = (c->prm.flags & FGPU_FLAG_RECORD_STOPS) != 0;
    StubStops sb;
    sb.seq = c->batch_seq++;
    std::vector<uint64_t> ext(4096);
    std::vector<uint32_t> info(4096);
    for (uint64_t i = 0; i < n_reads; i++) {
        const uint64_t len = offs[i + 1] - offs[i];
        uint64_t n = fo_scan_input_read_ex(c->sc, bases + offs[i], len, c->short_pf ? 0 : 1, ext.data(), info.data(), ext.size());
        if (n > ext.size()) return fail(c, FGPU_ERR_CAPACITY, "stub: more than 4096 list elements on one read");
        stub_paired_end(c, ext.data(), n);
        if (record)
            for (uint64_t e = 0; e < n; e++) {
                fgpu_stop s;
                s.ext = ext[e];
                s.read = (uint32_t)i;
                s.info = info[e];
                sb.stops.push_back(s);
            }
        c->scan_reads++;
    }
    if (record && (c->short_pf ? c->short_lists_to_host != 0 : c->long_mode == FGPU_LONG_PAIRS_OFF)) c->queue.push_back(sb);
    return FGPU_OK;
}

int fgpu_scan_batch(fgpu_ctx* c, const fgpu_reads* r) {
    if (!c || !r) return FGPU_ERR_ARG;
    if (c->phase != 2) return fail(c, FGPU_ERR_STATE, "scan_batch outside scan_begin/scan_end");
    const char* bases;
    const uint64_t* offs;
    flatten(c, r, &bases, &offs);
    return stub_scan_flat(c, bases, offs, r->n_reads);
}

// the pure stage has no counterpart in the oracle: a prepared batch is kept as it is and scanned when its turn comes
int fgpu_scan_prepare(fgpu_ctx* c, const fgpu_reads* r) {
    if (!c || !r) return FGPU_ERR_ARG;
    if (c->phase != 2) return fail(c, FGPU_ERR_STATE, "scan_prepare outside scan_begin/scan_end");
    const char* bases;
    const uint64_t* offs;
    flatten(c, r, &bases, &offs);
    c->prepared.emplace_back();
    fgpu_ctx::Prepared& p = c->prepared.back();
    p.n = r->n_reads;
    p.offs.assign(offs, offs + r->n_reads + 1);
    p.bases.assign(bases, bases + offs[r->n_reads]);
    if (p.bases.empty()) p.bases.push_back('N');
    return FGPU_OK;
}

int fgpu_scan_walk_prepared(fgpu_ctx* c) {
    if (!c) return FGPU_ERR_ARG;
    if (c->phase != 2) return fail(c, FGPU_ERR_STATE, "scan_walk_prepared outside scan_begin/scan_end");
    while (!c->prepared.empty()) {
        fgpu_ctx::Prepared& p = c->prepared.front();
        const int rc = stub_scan_flat(c, p.bases.data(), p.offs.data(), p.n);
        if (rc != FGPU_OK) return rc;
        c->prepared.pop_front();
    }
    return FGPU_OK;
}

int fgpu_scan_end(fgpu_ctx* c, fgpu_scan_stats* st) {
    if (!c) return FGPU_ERR_ARG;
    if (c->phase != 2) return fail(c, FGPU_ERR_STATE, "scan_end without scan_begin");
    if (st) {
        fo_scan_stats o;
        fo_scan_get_stats(c->sc, &o);
        memset(st, 0, sizeof(*st));
        st->reads_processed = c->scan_reads + c->carried_reads;
        st->unambiguous_reads = o.unambiguous_reads;
        st->reads_no_errors = o.reads_no_errors;
        st->nb_jcheck_kmer = o.nb_jcheck_kmer;
        st->nb_no_juncs = o.nb_no_juncs;
        st->nb_processed = o.nb_processed;
        st->nb_skipped = o.nb_skipped;
        st->n_junctions = o.n_junctions;
    }
    c->phase = 0;
    return FGPU_OK;
}

int fgpu_scan_take_stops(fgpu_ctx* c, fgpu_stop* out, uint64_t cap, uint64_t* n_out, int64_t* seq) {
    if (!c || !n_out || !seq) return FGPU_ERR_ARG;
    *n_out = 0;
    *seq = -1;
    // one batch behind a running scan, like the product: the newest batch stays with the "walk" until the next call or scan_end
    if (c->queue.empty() || (c->phase == 2 && c->queue.size() < 2)) return FGPU_OK;
    StubStops& sb = c->queue.front();
    *seq = sb.seq;
    *n_out = sb.stops.size();
    if (sb.stops.size() > cap || (!out && !sb.stops.empty())) return FGPU_ERR_CAPACITY;
    if (!sb.stops.empty()) memcpy(out, sb.stops.data(), sb.stops.size() * sizeof(fgpu_stop));
    c->queue.pop_front();
    return FGPU_OK;
}

int fgpu_scan_junction_count(fgpu_ctx* c, uint64_t* n) {
    if (!c || !n || !c->sc) return FGPU_ERR_ARG;
    *n = fo_scan_get_junctions(c->sc, 1, NULL, NULL, 0);
    return FGPU_OK;
}

int fgpu_scan_download_junctions(fgpu_ctx* c, uint64_t* keys, fgpu_junction* recs, uint64_t cap, uint64_t* n_out) {
    if (!c || !keys || !recs || !n_out || !c->sc) return FGPU_ERR_ARG;
    const uint64_t n = fo_scan_get_junctions(c->sc, 1, NULL, NULL, 0);
    if (n > cap) return fail(c, FGPU_ERR_CAPACITY, "junction buffers too small");
    static_assert(sizeof(fgpu_junction) == sizeof(fo_junction), "record layouts");
    fo_scan_get_junctions(c->sc, 1, keys, (fo_junction*)recs, cap);
    *n_out = n;
    return FGPU_OK;
}

// the reference's dump order: here from the container itself (the product computes it on the device from the caller's rehash schedule)
int fgpu_scan_dump_order(fgpu_ctx* c, const uint64_t* counts, const uint64_t* buckets, uint64_t n_rehashes, uint64_t n, uint32_t* order) {
    if (!c || !counts || !buckets || !n_rehashes || (n && !order) || !c->sc) return FGPU_ERR_ARG;
    const uint64_t have = fo_scan_get_junctions(c->sc, 1, NULL, NULL, 0);
    if (n > have) return fail(c, FGPU_ERR_STATE, "more keys than junctions");
    std::vector<uint64_t> keys(have ? have : 1);
    std::vector<fo_junction> recs(have ? have : 1);
    fo_scan_get_junctions(c->sc, 1, keys.data(), recs.data(), have);
    std::unordered_map<uint64_t, uint32_t> real;
    for (uint64_t i = 0; i < n; i++) real.insert(std::pair<uint64_t, uint32_t>(keys[i], (uint32_t)i));
    uint64_t at = 0;
    for (const auto& kv : real) order[at++] = kv.second;
    return FGPU_OK;
}

// ---- the rest of the ABI is not needed by the stub's callers ------------------------------------------------------------------------------
#define STUB_UNSUPPORTED(c) fail(c, FGPU_ERR_STATE, "not in the test stub (tests/stub/faucet_gpu_stub.cpp)")
int fgpu_load_fixup(fgpu_ctx* c, const void*, fgpu_load_stats*) { return STUB_UNSUPPORTED(c); }
int fgpu_load_fixup_state(fgpu_ctx* c, int* ready, uint64_t* budget) {     // (the stand-in keeps nothing resident: hosts take the presence protocol)
    if (!c) return FGPU_ERR_ARG;
    if (ready) *ready = 0;
    if (budget) *budget = 0;
    return FGPU_OK;
}
int fgpu_scan_set_eager(fgpu_ctx* c, int) { return c ? FGPU_OK : FGPU_ERR_ARG; }
int fgpu_profile_enable(fgpu_ctx* c, int) { return c ? FGPU_OK : FGPU_ERR_ARG; }
int fgpu_diag_host_waits(fgpu_ctx* c, uint64_t* waits, double* ms) { if (!c || !waits) return FGPU_ERR_ARG; *waits = 0; if (ms) *ms = 0; return FGPU_OK; }
int fgpu_scan_refresh_prepared(fgpu_ctx* c) { return c ? FGPU_OK : FGPU_ERR_ARG; }   // (the stub prepares nothing ahead: its walk sees the table as it stands)
int fgpu_diag_prepared_refresh(fgpu_ctx* c, uint64_t out[4]) { if (!c || !out) return FGPU_ERR_ARG; out[0] = out[1] = out[2] = out[3] = 0; return FGPU_OK; }
int fgpu_diag_sparse_link(fgpu_ctx* c, uint64_t out[2]) { if (!c || !out) return FGPU_ERR_ARG; out[0] = out[1] = 0; return FGPU_OK; }
int fgpu_diag_ovw_tables(fgpu_ctx* c, uint64_t* hw, uint64_t* cap) { if (!c || !hw || !cap) return FGPU_ERR_ARG; *hw = 0; *cap = 1ULL << 23; return FGPU_OK; }
int fgpu_diag_ovw(fgpu_ctx* c, uint64_t out[6]) { if (!c || !out) return FGPU_ERR_ARG; memset(out, 0, 6 * sizeof(uint64_t)); return FGPU_OK; }
int fgpu_stage3_set_junctions(fgpu_ctx* c, const uint64_t*, const fgpu_junction*, uint64_t) { return STUB_UNSUPPORTED(c); }
uint64_t fgpu_stage3_contig_words(int32_t k, int32_t max_read_length) { return (uint64_t)(2 * max_read_length + k + 31) / 32 + 1; }
int fgpu_stage3_find_neighbors(fgpu_ctx* c, const uint64_t*, const int8_t*, uint64_t, int32_t, fgpu_neighbor*, uint64_t*, uint64_t*, uint64_t) {
    return STUB_UNSUPPORTED(c);
}

// ---- read shards (faucet_amd/host/shard_host.h): the presence protocol of pass 1, the table hand-over of pass 2, the exchanges ------------------
// "Device" memory is host memory here, a stream is the calling thread: every call has finished when it returns.
int fgpu_presence_batch(fgpu_ctx* c, const fgpu_reads* r) {
    if (!c || !r) return FGPU_ERR_ARG;
    if (c->phase != 0) return fail(c, FGPU_ERR_STATE, "presence_batch while a pass is open");
    const char* bases;
    const uint64_t* offs;
    flatten(c, r, &bases, &offs);
    fo_load_stats st;
    fo_load_single_filter(c->b1, bases, offs, r->n_reads, c->prm.k, &st);      // every k-mer's bits, no order
    return FGPU_OK;
}
int fgpu_bloom_devptr(fgpu_ctx* c, int which, void** dptr, uint64_t* nbytes) {
    fo_bloom* b = c ? which_bloom(c, which) : NULL;
    if (!b || !dptr) return FGPU_ERR_ARG;
    *dptr = fo_bloom_bits(b);
    if (nbytes) *nbytes = fo_bloom_nbytes(b);
    return FGPU_OK;
}
int fgpu_bitmap_or(fgpu_ctx* c, void* dst, const void* src, uint64_t nbytes) {
    if (!c || !dst || !src) return FGPU_ERR_ARG;
    for (uint64_t i = 0; i < nbytes; i++) ((uint8_t*)dst)[i] |= ((const uint8_t*)src)[i];
    return FGPU_OK;
}
int fgpu_device_alloc(fgpu_ctx* c, uint64_t nbytes, void** dptr) { if (!c || !dptr) return FGPU_ERR_ARG; *dptr = calloc(nbytes ? nbytes : 16, 1); return *dptr ? FGPU_OK : FGPU_ERR_NOMEM; }
int fgpu_device_free(fgpu_ctx* c, void* dptr) { if (!c) return FGPU_ERR_ARG; free(dptr); return FGPU_OK; }
int fgpu_device_copy(fgpu_ctx* c, void* dst, const void* src, uint64_t nbytes) { if (!c) return FGPU_ERR_ARG; if (nbytes) memmove(dst, src, nbytes); return FGPU_OK; }
int fgpu_device_zero(fgpu_ctx* c, void* dst, uint64_t nbytes) { if (!c) return FGPU_ERR_ARG; if (nbytes) memset(dst, 0, nbytes); return FGPU_OK; }
int fgpu_scan_pairs_devptr(fgpu_ctx* c, int which, void** dptr, uint64_t* nbytes) {
    fo_bloom* b = !c ? NULL : which == 0 ? c->short_pf : which == 1 ? c->long_pf : NULL;
    if (!b || !dptr) return c ? fail(c, FGPU_ERR_STATE, "no such pair filter") : FGPU_ERR_ARG;
    *dptr = fo_bloom_bits(b);
    if (nbytes) *nbytes = fo_bloom_nbytes(b);
    return FGPU_OK;
}

// the junction table as it travels between shards: 32 bytes per record -- key, record, padding -- in creation order
struct StubEntry { uint64_t key; fo_junction rec; uint8_t pad[FGPU_TABLE_ENTRY_BYTES - 8 - sizeof(fo_junction)]; };
static_assert(sizeof(StubEntry) == FGPU_TABLE_ENTRY_BYTES, "table entry size");
int fgpu_scan_table_entries(fgpu_ctx* c, uint64_t* n) {
    if (!c || !n || !c->sc) return FGPU_ERR_ARG;
    *n = fo_scan_get_junctions(c->sc, 1, NULL, NULL, 0);
    return FGPU_OK;
}
int fgpu_scan_export_table(fgpu_ctx* c, void* buf, uint64_t buf_bytes, uint64_t* n_entries) {
    if (!c || !buf || !n_entries || !c->sc) return FGPU_ERR_ARG;
    const uint64_t n = fo_scan_get_junctions(c->sc, 1, NULL, NULL, 0);
    if (n * sizeof(StubEntry) > buf_bytes) return fail(c, FGPU_ERR_CAPACITY, "table buffer too small");
    std::vector<uint64_t> keys(n ? n : 1);
    std::vector<fo_junction> recs(n ? n : 1);
    fo_scan_get_junctions(c->sc, 1, keys.data(), recs.data(), n);
    StubEntry* e = (StubEntry*)buf;
    for (uint64_t i = 0; i < n; i++) { memset(&e[i], 0, sizeof(StubEntry)); e[i].key = keys[i]; e[i].rec = recs[i]; }
    *n_entries = n;
    return FGPU_OK;
}
int fgpu_scan_import_table(fgpu_ctx* c, const void* buf, uint64_t n, const fgpu_scan_stats* carried) {
    if (!c || (n && !buf) || !c->sc) return FGPU_ERR_ARG;
    if (c->phase != 2) return fail(c, FGPU_ERR_STATE, "import_table outside scan_begin/scan_end");
    std::vector<uint64_t> keys(n ? n : 1);
    std::vector<fo_junction> recs(n ? n : 1);
    const StubEntry* e = (const StubEntry*)buf;
    for (uint64_t i = 0; i < n; i++) { keys[i] = e[i].key; recs[i] = e[i].rec; }
    fo_scan_stats st;
    memset(&st, 0, sizeof(st));
    if (carried) {
        st.unambiguous_reads = carried->unambiguous_reads;
        st.reads_no_errors = carried->reads_no_errors;
        st.nb_jcheck_kmer = carried->nb_jcheck_kmer;
        st.nb_no_juncs = carried->nb_no_juncs;
        st.nb_processed = carried->nb_processed;
        st.nb_skipped = carried->nb_skipped;
        c->carried_reads = carried->reads_processed;
    }
    fo_scan_import(c->sc, keys.data(), recs.data(), n, carried ? &st : NULL);
    return FGPU_OK;
}
int fgpu_scan_import_hint(fgpu_ctx* c, const void*, uint64_t) { return c ? FGPU_OK : FGPU_ERR_ARG; }   // (the oracle has no preview to make)

// the group: mailboxes per (source, destination); a receive copies out of the sender's buffer, a send returns when that has happened
struct StubMsg { const void* ptr; uint64_t nbytes; bool copied; };
struct fgpu_group {
    int n;
    std::mutex m;
    std::condition_variable cv;
    std::vector<std::deque<StubMsg*> > chan;
    std::vector<std::vector<StubMsg*> > outstanding;
    std::vector<fgpu_ctx*> ctx;
    int attached, bar_count;
    uint64_t bar_gen;
    bool aborted;
    std::string err;
};
int fgpu_group_create(int n_ranks, int transport, fgpu_group** out) {
    if (!out || n_ranks < 1 || n_ranks > 64 || (transport != FGPU_TRANSPORT_COPY && transport != FGPU_TRANSPORT_RCCL)) return FGPU_ERR_ARG;
    fgpu_group* g = new fgpu_group();
    g->n = n_ranks;
    g->chan.resize((size_t)n_ranks * n_ranks);
    g->outstanding.resize((size_t)n_ranks);
    g->ctx.assign((size_t)n_ranks, NULL);
    g->attached = g->bar_count = 0;
    g->bar_gen = 0;
    g->aborted = false;
    *out = g;
    if (transport == FGPU_TRANSPORT_RCCL) { g->err = "the test stub has no RCCL transport"; return FGPU_ERR_HIP; }
    return FGPU_OK;
}
void fgpu_group_destroy(fgpu_group* g) {
    if (!g) return;
    std::vector<StubMsg*> left;
    for (auto& q : g->chan) for (StubMsg* msg : q) left.push_back(msg);
    for (auto& v : g->outstanding) for (StubMsg* msg : v) { bool seen = false; for (StubMsg* o : left) if (o == msg) seen = true; if (!seen) left.push_back(msg); }
    for (StubMsg* msg : left) delete msg;
    delete g;
}
const char* fgpu_group_last_error(const fgpu_group* g, int) { return g ? g->err.c_str() : "no group"; }
void fgpu_group_abort(fgpu_group* g) {
    if (!g) return;
    { std::lock_guard<std::mutex> lk(g->m); g->aborted = true; }
    g->cv.notify_all();
}
int fgpu_group_attach(fgpu_group* g, int rank, fgpu_ctx* c) {
    if (!g || rank < 0 || rank >= g->n || !c) return FGPU_ERR_ARG;
    std::unique_lock<std::mutex> lk(g->m);
    g->ctx[(size_t)rank] = c;
    if (++g->attached == g->n) g->cv.notify_all();
    else g->cv.wait(lk, [&] { return g->attached == g->n || g->aborted; });
    return g->aborted ? FGPU_ERR_STATE : FGPU_OK;
}
int fgpu_group_barrier(fgpu_group* g, int rank) {
    if (!g || rank < 0 || rank >= g->n) return FGPU_ERR_ARG;
    std::unique_lock<std::mutex> lk(g->m);
    const uint64_t gen = g->bar_gen;
    if (++g->bar_count == g->n) { g->bar_count = 0; g->bar_gen++; g->cv.notify_all(); return FGPU_OK; }
    g->cv.wait(lk, [&] { return g->aborted || g->bar_gen != gen; });
    return g->bar_gen != gen ? FGPU_OK : FGPU_ERR_STATE;
}
static StubMsg* stub_post(fgpu_group* g, int rank, int dst, const void* p, uint64_t nbytes) {
    StubMsg* msg = new StubMsg{p, nbytes, false};
    { std::lock_guard<std::mutex> lk(g->m); g->chan[(size_t)rank * g->n + dst].push_back(msg); }
    g->cv.notify_all();
    return msg;
}
static int stub_settle(fgpu_group* g, StubMsg* msg) {
    std::unique_lock<std::mutex> lk(g->m);
    g->cv.wait(lk, [&] { return g->aborted || msg->copied; });
    if (!msg->copied) return FGPU_ERR_STATE;
    lk.unlock();
    delete msg;
    return FGPU_OK;
}
int fgpu_group_recv(fgpu_group* g, int rank, int src, void* dev, uint64_t nbytes) {
    if (!g || rank < 0 || rank >= g->n || src < 0 || src >= g->n) return FGPU_ERR_ARG;
    std::unique_lock<std::mutex> lk(g->m);
    std::deque<StubMsg*>& q = g->chan[(size_t)src * g->n + rank];
    g->cv.wait(lk, [&] { return g->aborted || !q.empty(); });
    if (q.empty()) return FGPU_ERR_STATE;
    StubMsg* msg = q.front();
    if (msg->nbytes != nbytes) { g->err = "receive and send differ in size"; return FGPU_ERR_ARG; }
    q.pop_front();
    if (nbytes) memcpy(dev, msg->ptr, nbytes);       // (under the lock: the sender is still waiting, its buffer is valid)
    msg->copied = true;
    lk.unlock();
    g->cv.notify_all();
    return FGPU_OK;
}
int fgpu_group_send(fgpu_group* g, int rank, int dst, const void* dev, uint64_t nbytes) {
    if (!g || rank < 0 || rank >= g->n || dst < 0 || dst >= g->n || dst == rank) return FGPU_ERR_ARG;
    return stub_settle(g, stub_post(g, rank, dst, dev, nbytes));
}
int fgpu_group_send_async(fgpu_group* g, int rank, int dst, const void* dev, uint64_t nbytes) {
    if (!g || rank < 0 || rank >= g->n || dst < 0 || dst >= g->n || dst == rank) return FGPU_ERR_ARG;
    g->outstanding[(size_t)rank].push_back(stub_post(g, rank, dst, dev, nbytes));
    return FGPU_OK;
}
int fgpu_group_flush(fgpu_group* g, int rank) {
    if (!g || rank < 0 || rank >= g->n) return FGPU_ERR_ARG;
    std::vector<StubMsg*> out;
    out.swap(g->outstanding[(size_t)rank]);
    int rc = FGPU_OK;
    for (size_t i = 0; i < out.size(); i++) {
        if (rc == FGPU_OK) rc = stub_settle(g, out[i]);
        if (rc != FGPU_OK) g->outstanding[(size_t)rank].push_back(out[i]);      // (left to fgpu_group_destroy)
    }
    return rc;
}
int fgpu_group_probe(fgpu_group* g, int rank, int src, int* waiting, uint64_t* nbytes) {
    if (!g || rank < 0 || rank >= g->n || src < 0 || src >= g->n || !waiting) return FGPU_ERR_ARG;
    std::lock_guard<std::mutex> lk(g->m);
    std::deque<StubMsg*>& q = g->chan[(size_t)src * g->n + rank];
    *waiting = q.empty() ? 0 : 1;
    if (nbytes) *nbytes = q.empty() ? 0 : q.front()->nbytes;
    return g->aborted ? FGPU_ERR_STATE : FGPU_OK;
}
static void stub_slices(uint64_t nbytes, int n, std::vector<uint64_t>& lo, std::vector<uint64_t>& hi) {
    uint64_t step = (nbytes + (uint64_t)n - 1) / (uint64_t)n;
    step = (step + 15) & ~15ULL;
    lo.resize((size_t)n);
    hi.resize((size_t)n);
    for (int q = 0; q < n; q++) { lo[(size_t)q] = std::min<uint64_t>((uint64_t)q * step, nbytes); hi[(size_t)q] = std::min<uint64_t>((uint64_t)(q + 1) * step, nbytes); }
}
// the same slice schedule as the library's (faucet_amd/csrc/group.hip), byte arrays in host memory
int fgpu_group_or_allreduce(fgpu_group* g, int rank, void* bitmap, uint64_t nbytes) {
    if (!g || rank < 0 || rank >= g->n || !bitmap || (nbytes & 15)) return FGPU_ERR_ARG;
    if (g->n == 1) return FGPU_OK;
    std::vector<uint64_t> lo, hi;
    stub_slices(nbytes, g->n, lo, hi);
    uint8_t* bm = (uint8_t*)bitmap;
    const uint64_t mine = hi[(size_t)rank] - lo[(size_t)rank];
    std::vector<uint8_t> stage((size_t)(mine ? mine : 1) * (size_t)(g->n - 1));
    std::vector<StubMsg*> posted;
    for (int q = 0; q < g->n; q++) if (q != rank && hi[(size_t)q] > lo[(size_t)q]) posted.push_back(stub_post(g, rank, q, bm + lo[(size_t)q], hi[(size_t)q] - lo[(size_t)q]));
    int i = 0, rc = FGPU_OK;
    for (int q = 0; q < g->n && rc == FGPU_OK; q++) { if (q == rank) continue; if (mine) rc = fgpu_group_recv(g, rank, q, stage.data() + (size_t)i * mine, mine); i++; }
    for (StubMsg* msg : posted) { const int r2 = stub_settle(g, msg); if (rc == FGPU_OK) rc = r2; }
    if (rc != FGPU_OK) return rc;
    for (int s2 = 0; mine && s2 < g->n - 1; s2++) for (uint64_t b = 0; b < mine; b++) bm[lo[(size_t)rank] + b] |= stage[(size_t)s2 * mine + b];
    posted.clear();
    for (int q = 0; q < g->n; q++) if (q != rank && mine) posted.push_back(stub_post(g, rank, q, bm + lo[(size_t)rank], mine));
    for (int q = 0; q < g->n && rc == FGPU_OK; q++) if (q != rank && hi[(size_t)q] > lo[(size_t)q]) rc = fgpu_group_recv(g, rank, q, bm + lo[(size_t)q], hi[(size_t)q] - lo[(size_t)q]);
    for (StubMsg* msg : posted) { const int r2 = stub_settle(g, msg); if (rc == FGPU_OK) rc = r2; }
    return rc;
}
int fgpu_group_exclusive_prefix_or(fgpu_group* g, int rank, const void* bitmap, void* out_v, uint64_t nbytes) {
    if (!g || rank < 0 || rank >= g->n || !bitmap || !out_v || (nbytes & 15)) return FGPU_ERR_ARG;
    uint8_t* out = (uint8_t*)out_v;
    if (g->n == 1) { memset(out, 0, nbytes); return FGPU_OK; }
    std::vector<uint64_t> lo, hi;
    stub_slices(nbytes, g->n, lo, hi);
    const uint8_t* bm = (const uint8_t*)bitmap;
    const uint64_t mine = hi[(size_t)rank] - lo[(size_t)rank], cell = mine ? mine : 1;
    std::vector<uint8_t> stage((size_t)cell * (size_t)g->n), pref((size_t)cell * (size_t)g->n, 0);
    if (mine) memcpy(stage.data() + (size_t)rank * mine, bm + lo[(size_t)rank], mine);
    std::vector<StubMsg*> posted;
    for (int q = 0; q < g->n; q++) if (q != rank && hi[(size_t)q] > lo[(size_t)q]) posted.push_back(stub_post(g, rank, q, bm + lo[(size_t)q], hi[(size_t)q] - lo[(size_t)q]));
    int rc = FGPU_OK;
    for (int q = 0; q < g->n && rc == FGPU_OK; q++) if (q != rank && mine) rc = fgpu_group_recv(g, rank, q, stage.data() + (size_t)q * mine, mine);
    for (StubMsg* msg : posted) { const int r2 = stub_settle(g, msg); if (rc == FGPU_OK) rc = r2; }
    if (rc != FGPU_OK) return rc;
    for (int q = 1; mine && q < g->n; q++) for (uint64_t b = 0; b < mine; b++) pref[(size_t)q * mine + b] = pref[(size_t)(q - 1) * mine + b] | stage[(size_t)(q - 1) * mine + b];
    posted.clear();
    for (int q = 1; q < g->n; q++) if (q != rank && mine) posted.push_back(stub_post(g, rank, q, pref.data() + (size_t)q * mine, mine));
    if (rank != 0) for (int q = 0; q < g->n && rc == FGPU_OK; q++) if (q != rank && hi[(size_t)q] > lo[(size_t)q]) rc = fgpu_group_recv(g, rank, q, out + lo[(size_t)q], hi[(size_t)q] - lo[(size_t)q]);
    for (StubMsg* msg : posted) { const int r2 = stub_settle(g, msg); if (rc == FGPU_OK) rc = r2; }
    if (rc != FGPU_OK) return rc;
    if (rank == 0) memset(out, 0, nbytes);
    else if (mine) memcpy(out + lo[(size_t)rank], pref.data() + (size_t)rank * mine, mine);
    return FGPU_OK;
}
int fgpu_group_selftest(fgpu_group* g, int rank, uint64_t nbytes, int* ok) { if (!g || rank < 0 || rank >= g->n || !ok || !nbytes) return FGPU_ERR_ARG; *ok = 1; return FGPU_OK; }

}  // extern "C"
