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

#include <deque>
#include <string>
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
};

static std::string g_create_error;

static int fail(fgpu_ctx* c, int rc, const char* what) {
    if (c) c->err = what; else g_create_error = what;
    return rc;
}

extern "C" {

int fgpu_abi_version(void) { return FGPU_ABI_VERSION; }
int fgpu_device_count(void) { return 0; }

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
    c->phase = 2;
    return FGPU_OK;
}

int fgpu_scan_batch(fgpu_ctx* c, const fgpu_reads* r) {
    if (!c || !r) return FGPU_ERR_ARG;
    if (c->phase != 2) return fail(c, FGPU_ERR_STATE, "scan_batch outside scan_begin/scan_end");
    const char* bases;
    const uint64_t* offs;
    flatten(c, r, &bases, &offs);
    const bool record = (c->prm.flags & FGPU_FLAG_RECORD_STOPS) != 0;
    StubStops sb;
    sb.seq = c->batch_seq++;
    std::vector<uint64_t> ext(4096);
    std::vector<uint32_t> info(4096);
    for (uint64_t i = 0; i < r->n_reads; i++) {
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

int fgpu_scan_end(fgpu_ctx* c, fgpu_scan_stats* st) {
    if (!c) return FGPU_ERR_ARG;
    if (c->phase != 2) return fail(c, FGPU_ERR_STATE, "scan_end without scan_begin");
    if (st) {
        fo_scan_stats o;
        fo_scan_get_stats(c->sc, &o);
        memset(st, 0, sizeof(*st));
        st->reads_processed = c->scan_reads;
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

// ---- the rest of the ABI is not needed by the stub's callers ------------------------------------------------------------------------------
#define STUB_UNSUPPORTED(c) fail(c, FGPU_ERR_STATE, "not in the test stub (tests/stub/faucet_gpu_stub.cpp)")
int fgpu_presence_batch(fgpu_ctx* c, const fgpu_reads*) { return STUB_UNSUPPORTED(c); }
int fgpu_load_fixup(fgpu_ctx* c, const void*, fgpu_load_stats*) { return STUB_UNSUPPORTED(c); }
int fgpu_bloom_devptr(fgpu_ctx* c, int, void**, uint64_t*) { return STUB_UNSUPPORTED(c); }
int fgpu_bitmap_or(fgpu_ctx* c, void*, const void*, uint64_t) { return STUB_UNSUPPORTED(c); }
int fgpu_scan_prepare(fgpu_ctx* c, const fgpu_reads*) { return STUB_UNSUPPORTED(c); }
int fgpu_scan_walk_prepared(fgpu_ctx* c) { return STUB_UNSUPPORTED(c); }
int fgpu_scan_set_eager(fgpu_ctx* c, int) { return c ? FGPU_OK : FGPU_ERR_ARG; }
int fgpu_profile_enable(fgpu_ctx* c, int) { return c ? FGPU_OK : FGPU_ERR_ARG; }
int fgpu_diag_ovw(fgpu_ctx* c, uint64_t out[6]) { if (!c || !out) return FGPU_ERR_ARG; memset(out, 0, 6 * sizeof(uint64_t)); return FGPU_OK; }
int fgpu_scan_table_entries(fgpu_ctx* c, uint64_t*) { return STUB_UNSUPPORTED(c); }
int fgpu_scan_export_table(fgpu_ctx* c, void*, uint64_t, uint64_t*) { return STUB_UNSUPPORTED(c); }
int fgpu_scan_import_table(fgpu_ctx* c, const void*, uint64_t, const fgpu_scan_stats*) { return STUB_UNSUPPORTED(c); }
int fgpu_scan_import_hint(fgpu_ctx* c, const void*, uint64_t) { return STUB_UNSUPPORTED(c); }
int fgpu_stage3_set_junctions(fgpu_ctx* c, const uint64_t*, const fgpu_junction*, uint64_t) { return STUB_UNSUPPORTED(c); }
uint64_t fgpu_stage3_contig_words(int32_t k, int32_t max_read_length) { return (uint64_t)(2 * max_read_length + k + 31) / 32 + 1; }
int fgpu_stage3_find_neighbors(fgpu_ctx* c, const uint64_t*, const int8_t*, uint64_t, int32_t, fgpu_neighbor*, uint64_t*, uint64_t*, uint64_t) {
    return STUB_UNSUPPORTED(c);
}

}  // extern "C"
