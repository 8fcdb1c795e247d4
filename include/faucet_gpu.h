/*
 * faucet_gpu.h — C ABI of libfaucet_gpu.so: Faucet's two-pass k-mer pipeline on MI355X (gfx950).
 *
 * The reference (Shamir-Lab/Faucet) has no plugin / FFI layer; this ABI cuts the two seams that
 * SURVEY.md §8(b) names and is what a maintainer's `faucet` binary (or any other host language)
 * binds.  Every entry point cites the reference interface it replaces.
 *
 *   pass 1   void load_two_filters(Bloom*, Bloom*, std::string, bool fastq, bool mercy)
 *            utils/Bloom.h:294, utils/Bloom.cpp:267, caller src/Faucet.cpp:220
 *   pass 2   void ReadScanner::scanReads(bool fastq, bool paired_ends, bool no_cleaning)
 *            src/ReadScanner.cpp:284, constructed and called at src/Faucet.cpp:241-245
 *   sizing   getBloomFilterFromReads / create_bloom_filter_optimal / _2_hash / Bloom::Bloom
 *            src/Faucet.cpp:197-219, utils/Bloom.cpp:165-247
 *
 * Conventions: extern "C", opaque context, POD structs, caller-owned buffers with explicit sizes,
 * int status (0 = ok), no exceptions / STL / torch types across the boundary.  One host thread
 * drives one context; one context drives one HIP device.  Results are bit-identical to the
 * reference CPU path (same Bloom bit array, same junction records).  There is no CPU fallback:
 * every compute entry point returns FGPU_ERR_HIP when no gfx950 device is usable.
 */
#ifndef FAUCET_GPU_H
#define FAUCET_GPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FGPU_ABI_VERSION 3

enum {
    FGPU_OK = 0,
    FGPU_ERR_ARG = 1,       /* bad parameter (k out of 1..31, tai not a power of two, ...) */
    FGPU_ERR_HIP = 2,       /* HIP runtime error or no device; see fgpu_last_error() */
    FGPU_ERR_STATE = 3,     /* call out of sequence (e.g. load_batch before load_begin) */
    FGPU_ERR_CAPACITY = 4,  /* batch larger than max_batch_bases, junction table full, ... */
    FGPU_ERR_NOMEM = 5
};

enum { FGPU_BLOO1 = 0, FGPU_BLOO2 = 1 };

typedef struct fgpu_ctx fgpu_ctx;

typedef struct {
    int32_t  k;                 /* -size_kmer (src/Faucet.cpp:66), 1..31 */
    int32_t  j;                 /* -j, default 1 (src/Faucet.h:15) */
    int32_t  max_spacer_dist;   /* -max_spacer_dist, default 100 (src/Faucet.h:48) */
    int32_t  n_hash;            /* hash functions of both load filters (utils/Bloom.cpp:243-244) */
    uint64_t tai;               /* bits per load filter, power of two (utils/Bloom.cpp:173-175) */
    int32_t  device;            /* HIP device ordinal */
    int32_t  flags;             /* FGPU_FLAG_* */
    uint64_t junction_capacity; /* INITIAL slots of the device junction table, power of two; 0 = default (tai / 32).  The table is
                                 * rehashed into a larger one between batches whenever it is more than a quarter full (the
                                 * reference's unordered_map grows the same way, utils/JunctionMap.h:61).  A single batch that outgrows
                                 * it is absorbed by the library while the scan's batches are in its journal (see fgpu_scan_set_eager:
                                 * the journal is scanned again on a table four times the size; fgpu_diag_scan_replays counts it); only a
                                 * scan without a journal -- FGPU_FLAG_EAGER_FLAGS, fgpu_scan_set_eager(1), or beyond the journal's
                                 * budget of an eighth of the device memory -- ends with FGPU_ERR_CAPACITY there. */
    uint64_t max_batch_bases;   /* largest batch (bases + one separator per read); 0 = default 2^30 */
    void*    stream;            /* hipStream_t to run on, or NULL for a private stream */
    uint64_t walk_window_span;  /* stream positions per scheduling window of the ordered walk; 0 = adaptive.
                                 * Any value gives the same results (tests use small ones to force clusters). */
} fgpu_params;

#define FGPU_FLAG_PROFILE 1     /* bracket every kernel with HIP events (fgpu_kernel_times) */
#define FGPU_FLAG_MERCY 16       /* load_two_filters(..., mercy = true): also add low-coverage k-mers between solid ones to bloo2
                                 * (utils/Bloom.cpp:300-333; the reference's --mercy) */
#define FGPU_FLAG_RECORD_STOPS 8 /* keep scanInputRead's return value for every read (fgpu_scan_take_stops) */
#define FGPU_FLAG_NO_RESIDENT 4 /* do not keep the load batches in HBM for the scan pass (see fgpu_load_batch) */
#define FGPU_FLAG_KEY_ORDER_FROM_START 32 /* large clusters are walked in the order of their junction k-mers' turns from a scan's first window on,
                                 * not only once the scan has shown such a cluster, and "large" means 32 pieces instead of 64 (DESIGN.md
                                 * section 4): four small launches more per window; same results.  For callers that expect repeats at
                                 * high coverage. */
#define FGPU_FLAG_EAGER_FLAGS 2 /* evaluate testForJunction at every position instead of only where the walk can stop
                                 * skipping (same results; the lazy default checks itself and repairs what it finds, see fgpu_scan_set_eager) */

/* A batch of sequence lines in file order: read i = bases[offsets[i] .. offsets[i+1]).  Any byte
 * may occur; everything except upper-case A C G T splits a read exactly as isValidNuc /
 * getUnambiguousReads do (utils/Kmer.cpp:50-80).  offsets[0] need not be 0.
 * on_device != 0: all pointers are device (HBM) pointers on params.device.
 * starts != NULL (device batches only): the reads are not contiguous in `bases` -- read i is
 * bases[starts[i] .. starts[i] + offsets[i+1] - offsets[i]) (sequence lines inside raw FASTA/FASTQ text,
 * see fgpu_text_split); offsets still carries the lengths as differences.
 * total_bases (device batches only; host batches: ignored): offsets[n_reads] - offsets[0] if the caller knows it, else 0.  With 0 the library
 * reads the two offsets back from the device, which makes every call wait for everything it has queued before -- the kernels of the batch
 * before -- before it can queue this batch's: a caller that streams device batches should fill it in (round 4, config 2: the host spent 9 ms of a
 * 64 ms scan in that wait).  A value that does not match the offsets is an error the pass reports at its next synchronising call (FGPU_ERR_ARG). */
typedef struct {
    const char*     bases;
    const uint64_t* offsets;    /* n_reads + 1 entries */
    uint64_t        n_reads;
    int32_t         on_device;
    uint32_t        total_bases;
    const uint64_t* starts;     /* n_reads entries, or NULL */
} fgpu_reads;

typedef struct {
    uint64_t reads_processed;    /* "Reads processed"   utils/Bloom.cpp:335,346 */
    uint64_t unambiguous_reads;  /* "Unambiguous reads" utils/Bloom.cpp:287,347 */
    uint64_t kmers;              /* iterations of the loop at utils/Bloom.cpp:289 — the unit N */
    uint64_t to_bloo2;           /* occurrences routed to bloo2 (utils/Bloom.cpp:293) */
} fgpu_load_stats;

typedef struct {
    uint64_t reads_processed;    /* src/ReadScanner.cpp:348 */
    uint64_t unambiguous_reads;  /* :269 */
    uint64_t reads_no_errors;    /* :276 */
    uint64_t nb_jcheck_kmer;     /* :49  */
    uint64_t nb_no_juncs;        /* :196 */
    uint64_t nb_processed;       /* :83,:192 */
    uint64_t nb_skipped;         /* :192 */
    uint64_t n_junctions;        /* JunctionMap::getNumJunctions, utils/JunctionMap.cpp:598-600 */
    uint64_t kmers;              /* windows inside unambiguous segments (same unit N as pass 1) */
    /* diagnostics of the ordered walk (not in the reference) */
    uint64_t walk_windows;       /* scheduling windows executed */
    uint64_t walk_followers;     /* pieces that had to wait for an earlier piece of their cluster */
    uint64_t walk_max_cluster;   /* largest dependency cluster seen */
    uint64_t flag_positions;     /* window positions at which testForJunction was evaluated (lazy flags: a subset of all) */
    uint64_t piece_positions;    /* window positions inside valid pieces */
    uint64_t valid_reused;       /* getValidReads answers taken from the load pass' resident planes (no filter probe) */
    uint64_t flags_filled;       /* windows whose testForJunction the walk evaluated itself because the preview had left them out */
    uint64_t walk_parallel;      /* pieces of large clusters walked apart from their cluster's thread: in the order of their junction k-mers'
                                  * turns (k_walk_ko, the default for clusters of 64 pieces and more) or out of order (FGPU_WALK_HEAVY) */
} fgpu_scan_stats;

/* One element of the list ReadScanner::scanInputRead returns for a read (src/ReadScanner.cpp:260-282): the real-extension
 * k-mer pushed at a junction visit of scan_forward (:140) or add_fake_junction's extension (:197,:103), with what the
 * callers of that list need to know about the visit (:147-169, :208-225).  Needs FGPU_FLAG_RECORD_STOPS. */
typedef struct {
    uint64_t ext;    /* ReadKmer::getRealExtension() at the visit */
    uint32_t read;   /* index of the read within its batch */
    uint32_t info;   /* bits 0-27 ReadKmer::pos inside the valid piece; FGPU_STOP_* bits */
} fgpu_stop;
#define FGPU_STOP_POS_MASK 0x0FFFFFFFu
#define FGPU_STOP_FORWARD  (1u << 28)   /* facing FORWARD (else BACKWARD) */
#define FGPU_STOP_FIRST    (1u << 29)   /* first element of its valid piece (= of one scan_forward call) */
#define FGPU_STOP_FAKE     (1u << 30)   /* the piece had no junction: this is add_fake_junction's element */

/* Junction record, field for field utils/Junction.h:10-18 (cov is private there). */
typedef struct {
    uint8_t cov[4];
    uint8_t dist[5];
    uint8_t linked[5];
} fgpu_junction;

/* ---- library / context ------------------------------------------------------------------- */
int         fgpu_abi_version(void);
/* Number of usable gfx950 devices (0 when there is none; never fails). */
int         fgpu_device_count(void);
int         fgpu_create(const fgpu_params* params, fgpu_ctx** out);
void        fgpu_destroy(fgpu_ctx* ctx);
const char* fgpu_last_error(const fgpu_ctx* ctx);   /* ctx may be NULL: last fgpu_create failure */
int         fgpu_synchronize(fgpu_ctx* ctx);

/* ---- filter sizing (host arithmetic only; replaces src/Faucet.cpp:197-219, Bloom.cpp:165-247) --- */
/* p1 such that two filters at rate p1 give overall rate fp; <0 when the root is not bracketed
 * (e.g. singletons == 0) — callers must treat that as an argument error. */
double   fgpu_solve_p1(uint64_t estimated_kmers, uint64_t singletons, float fp, int32_t* iterations);
uint64_t fgpu_bloom_tai(uint64_t requested_bits);                                  /* Bloom.cpp:173-178 */
void     fgpu_size_optimal(uint64_t estimated, float fp, int32_t* bits_per_item, uint64_t* tai, int32_t* n_hash);
void     fgpu_size_two_hash(uint64_t estimated, float fp, int32_t* bits_per_item, uint64_t* tai, int32_t* n_hash);

/* ---- pass 1: Bloom load (replaces load_two_filters, utils/Bloom.cpp:267-350) ------------------- */
/* Zero both filters (or keep bloo1's current content as the carried-in state when keep_carry != 0:
 * multi-GPU shards start from the prefix-OR of the lower ranks' k-mer presence bitmaps). */
int fgpu_load_begin(fgpu_ctx* ctx, int keep_carry);
#define FGPU_LOAD_KEEP_CARRY   1   /* bloo1's current content is the carried-in state */
#define FGPU_LOAD_SHARD_TIMES  2   /* the pass of a read shard that fgpu_load_fixup will complete, for shards of fewer than 2^32 - 16 stream positions
                                    * (FGPU_ERR_CAPACITY beyond): first-set times count from the start of the pass, not of an epoch, and the fix-up
                                    * compares them.  Costs the pass nothing. */
#define FGPU_LOAD_SHARD_PLANES 4   /* the same for shards of ANY size (round 5): while it resolves its occurrences the pass writes down which of an
                                    * occurrence's bits were NOT set before it within the shard (n_hash <= 4 planes of one bit per position, kept in HBM
                                    * with the batch), and the fix-up asks the lower shards' bits for exactly those.  The resolve kernel then looks at every
                                    * missing bit instead of stopping at the first that fails (+10 % on the pass). */
/* Consume one batch, in file order.  Exact: occurrence t goes to bloo2 iff all its bits were set
 * by occurrences < t, t following the reference's processing order. */
int fgpu_load_batch(fgpu_ctx* ctx, const fgpu_reads* reads);
int fgpu_load_end(fgpu_ctx* ctx, fgpu_load_stats* stats);
/* OR the bits of every k-mer of the batch into bloo1 with no ordering (presence bitmap; used by
 * multi-GPU shards before the prefix-OR exchange).  Same unit counters as load_batch. */
int fgpu_presence_batch(fgpu_ctx* ctx, const fgpu_reads* reads);
/* Read shards without the presence pass.  After a load pass of THIS shard alone (begun with FGPU_LOAD_SHARD_TIMES, or with
 * FGPU_LOAD_SHARD_PLANES and at most 4 hash functions; an empty carry, all batches kept resident, no --mercy) and the exchange of the shards' bloo1, `prefix_dev` = OR of the bloo1
 * of all lower shards (device pointer, tai/8 bytes) is what the sequential run has in bloo1 when it reaches this shard.  Every occurrence the local
 * pass kept out of bloo2 is looked at again: it goes to bloo2 iff each of its bits is in the prefix or was set locally before it (by the pass's
 * times, or by the planes it wrote down).  Adds
 * those occurrences to bloo2, to the planes the scan reuses and to stats->to_bloo2; bloo1 |= prefix.  FGPU_ERR_STATE when the pass was not
 * begun that way or a batch was not kept. */
int fgpu_load_fixup(fgpu_ctx* ctx, const void* prefix_dev, fgpu_load_stats* stats);
/* What a host needs to CHOOSE between the two protocols of pass 1 and to fall back from one to the other (shard_host.h had a
 * constant of its own and no way back).  *ready (may be NULL): 1 iff fgpu_load_fixup can complete the load pass that has just ended -- 0 when
 * a batch was not kept resident (budget, or no memory at that moment), the pass was not begun as a shard's, or --mercy: the ranks then
 * agree to run the presence protocol instead (every rank the same).  *resident_budget_bytes (may be NULL): device bytes this context keeps
 * load batches resident in; a shard's batches need about one byte per base (codes, bad, sure, and up to four fail planes). */
int fgpu_load_fixup_state(fgpu_ctx* ctx, int* ready, uint64_t* resident_budget_bytes);

/* filters: raw bit arrays, tai/8 bytes, exactly the .bloom file body (utils/Bloom.cpp:571-587) */
int fgpu_bloom_download(fgpu_ctx* ctx, int which, uint8_t* host_out, uint64_t nbytes);
/* The same copy, started now and finished by fgpu_bloom_download_wait: it runs on its own copy stream behind the work
 * submitted so far, next to whatever the caller submits afterwards (the reference's bloo2 stays in host memory for the dump
 * and for Stage 3, src/Faucet.cpp:220-245 -- the scan in between does not read the host copy).  host_out should be page-locked
 * (fgpu_host_alloc); with pageable memory the call simply blocks until the copy is done.  The filter must not be rewritten
 * (load_begin, bloom_upload) before the wait: those calls wait themselves.  One download in flight per context. */
int fgpu_bloom_download_begin(fgpu_ctx* ctx, int which, uint8_t* host_out, uint64_t nbytes);
int fgpu_bloom_download_wait(fgpu_ctx* ctx);
int fgpu_bloom_upload(fgpu_ctx* ctx, int which, const uint8_t* host_in, uint64_t nbytes);   /* -bloom_file restart */
int fgpu_bloom_weight(fgpu_ctx* ctx, int which, float* weight);                             /* Bloom::weight, Bloom.cpp:191-203 */
int fgpu_bloom_devptr(fgpu_ctx* ctx, int which, void** dptr, uint64_t* nbytes);
/* dst |= src over nbytes (device pointers, nbytes multiple of 16): the local step of the
 * prefix-OR / OR-allreduce that RCCL cannot express as a reduction op. */
int fgpu_bitmap_or(fgpu_ctx* ctx, void* dst_dev, const void* src_dev, uint64_t nbytes);

/* ---- input side: record splitting on the device ------------------------------------------------
 * The reference's reading loop (utils/Bloom.cpp:280-282,340; src/ReadScanner.cpp:306-308,349):
 *     while (getline(header)) { getline(sequence); ...; if (fastq) { getline; getline; } }
 * applied to `nbytes` of file text (host or device memory).  Fills *out with a DEVICE batch (bases = the text on the
 * device, starts/offsets = the sequence line of every record) that stays valid until the next fgpu_text_split on this
 * context, ready for fgpu_load_batch / fgpu_scan_batch / fgpu_presence_batch.
 * final_chunk == 0: only records whose 2 (FASTA) or 4 (FASTQ) lines all end inside the text are taken; *consumed = bytes
 * they occupy -- the caller prepends the rest to the next chunk.  final_chunk != 0: the text ends the file; a last line
 * without a newline, a missing sequence line (empty read) and missing FASTQ tail lines are handled as getline handles
 * them; *consumed = nbytes.  A carriage return before the newline stays part of the line, as it does in the reference. */
/* Page-locked host memory for the text handed to fgpu_text_split (or for read batches): copies to the device then run at
 * link speed and asynchronously.  NULL when it cannot be had; plain malloc'ed memory works too, only slower. */
void* fgpu_host_alloc(uint64_t bytes);
void  fgpu_host_free(void* p);
/* Announces the largest chunk of text (bytes) the caller will hand to fgpu_text_split, so that the first, smaller chunks of a pass do not each
 * make the library free and re-allocate its working buffers (every re-allocation synchronises the device).  A hint: larger chunks still work. */
int fgpu_text_reserve(fgpu_ctx* ctx, uint64_t max_chunk_bytes);
int fgpu_text_split(fgpu_ctx* ctx, const char* text, uint64_t nbytes, int text_on_device, int fastq, int final_chunk,
                    fgpu_reads* out, uint64_t* consumed);

/* ---- pass 2: junction scan (replaces ReadScanner::scanReads, src/ReadScanner.cpp:284-359) ------- */
/* Uses bloo2 as resident on the device (after fgpu_load_end or fgpu_bloom_upload).
 * Device memory: the first scan of a context allocates the junction table (56 bytes per slot of junction_capacity) and the walk's window
 * tables -- 40 bytes per position of the largest scheduling window: 2.5 GiB for filters up to 2^30 bits, up to 10 GiB (windows of 2^28
 * positions) for filters of 2^32 bits and more.  The larger windows only buy speed on thin coverage: where a quarter of the free device
 * memory does not hold them (several contexts on one device, a smaller device) the bound is halved, down to 2^26 positions, before
 * FGPU_ERR_NOMEM is returned. */
int fgpu_scan_begin(fgpu_ctx* ctx);
/* Pure stage + ordered walk for one batch, in file order. */
int fgpu_scan_batch(fgpu_ctx* ctx, const fgpu_reads* reads);
/* The same in two steps, so that read shards can do the pure part concurrently and only the
 * ordered walk is serial: scan_prepare runs the pure stage and keeps its bit planes resident
 * (about 1.6 bytes of HBM per base); scan_walk_prepared walks every prepared batch in the order
 * they were prepared.  scan_batch(b) == scan_prepare(b); scan_walk_prepared(). */
int fgpu_scan_prepare(fgpu_ctx* ctx, const fgpu_reads* reads);
int fgpu_scan_walk_prepared(fgpu_ctx* ctx);
int fgpu_scan_end(fgpu_ctx* ctx, fgpu_scan_stats* stats);
/* The scan evaluates testForJunction only where its preview of the walk says the walk can stop (lazy
 * flags); where the walk scans a window outside that preview it evaluates the tests itself, so the results are exact either
 * way.  The one case the walk cannot absorb on the spot is such a late test coming out TRUE (a junction at a k-mer the window's dependency
 * clusters did not know, AND whose k-mer occurs on another piece of the same window).  The library absorbs that too: while a scan is lazy it keeps the packed form of every batch in HBM (3 bits per
 * base, up to an eighth of the device memory) and, should the case arise, resets the junction map and scans those batches again by itself
 * with every test evaluated -- the caller sees an ordinary scan and never has to hand its reads over twice (they may come from a pipe).
 * A scan that outgrows that journal is verified up to there and goes on with eager tests.  fgpu_diag_scan_replays counts the replays;
 * fgpu_scan_stats.flags_filled counts the windows evaluated inside the walk.  fgpu_scan_set_eager(ctx, 1) switches the preview off for the
 * following scans altogether (same results as FGPU_FLAG_EAGER_FLAGS, about 1.5x the junction-test probes, no journal). */
int fgpu_scan_set_eager(fgpu_ctx* ctx, int on);
/* The short pair filter on the device.  With cleaning on, scan_forward feeds every piece's result list to
 * short_pair_filter->addPair (src/ReadScanner.cpp:208-225; Bloom::addPair utils/Bloom.cpp:127-139): adds only, so their order is free.
 * After fgpu_scan_short_pairs(tai, n_hash) -- the filter create_bloom_filter_optimal would make, src/Faucet.cpp:266-283 -- every scan keeps
 * that filter in HBM and applies the rules to each batch's lists as they are harvested; fgpu_scan_short_pairs_download (after
 * fgpu_scan_end) returns its tai / 8 bytes, the content of the reference's .short_pair_filter.  lists_to_host = 0: the lists are not
 * brought to the host at all (nothing else reads them, or their other reader -- the long pair filter, fgpu_scan_long_pairs -- is on the
 * device as well; fgpu_scan_take_stops then reports no batches); != 0: they are still handed out (a caller that applies the paired-end
 * loop itself).
 * Needs FGPU_FLAG_RECORD_STOPS; tai = 0 switches it off again.  Only between passes. */
int fgpu_scan_short_pairs(fgpu_ctx* ctx, uint64_t tai, int32_t n_hash, int32_t lists_to_host);
int fgpu_scan_short_pairs_download(fgpu_ctx* ctx, uint8_t* out, uint64_t n_bytes);

/* The long pair filter on the device (src/ReadScanner.cpp:317-343 with Bloom::containsPair / addPair, utils/Bloom.cpp:127-154).
 * With --paired_ends the reference treats consecutive records of the scan file as the two ends of a pair and, when both of their
 * scanInputRead lists are non-empty, CHECKS every k-mer of the first end's list for a partner among the second end's
 * (long_pair_filter->containsPair) and INSERTS (k-mer, second end's first k-mer) when there is none -- check-then-insert in file order.
 * After fgpu_scan_long_pairs(tai, n_hash, FGPU_LONG_PAIRS_FILTER) -- the filter create_bloom_filter_optimal would make, src/Faucet.cpp:268-281 --
 * every scan keeps that filter in HBM and applies the loop to each batch's lists as they are harvested, exactly: every list element gets
 * its file-order time, inserts post first-set times per filter bit, and the check / insert decisions are iterated to their fixed point,
 * which is the sequential run's (faucet_amd/csrc/pairs.hip; 4 bytes of HBM per filter bit beside the filter).  Reads 2p and 2p+1 of the
 * SCAN (counted over all its batches, empty records included) are a pair; a first end at the end of a batch waits for the next batch.
 * FGPU_LONG_PAIRS_COUNT: only the reference's "Empty count / not empty count" (what --no_cleaning leaves of the loop); tai, n_hash unused.
 * fgpu_scan_long_pairs_download (after fgpu_scan_end): the tai / 8 bytes of the reference's .long_pair_filter (out may be NULL) and the two
 * counts.  Lists then leave the device only if fgpu_scan_short_pairs(..., lists_to_host != 0) asks for them.
 * Needs FGPU_FLAG_RECORD_STOPS; FGPU_LONG_PAIRS_OFF switches it off again.  Only between passes. */
#define FGPU_LONG_PAIRS_OFF 0
#define FGPU_LONG_PAIRS_COUNT 1
#define FGPU_LONG_PAIRS_FILTER 2
int fgpu_scan_long_pairs(fgpu_ctx* ctx, uint64_t tai, int32_t n_hash, int32_t mode);
int fgpu_scan_long_pairs_download(fgpu_ctx* ctx, uint8_t* out, uint64_t n_bytes, uint64_t* empty_count, uint64_t* not_empty_count);

/* scanInputRead's lists of one scanned batch, flattened in processing order (reads in file order; inside a read the
 * valid pieces in the order scanInputRead walks them; inside a piece by half-step).  Batches come out in scan order,
 * one per call: *batch_seq = number of the batch within the scan, or -1 (and *n_out = 0) when none is left.  The
 * call waits for the ordered walk of that batch only, so calling it once after every fgpu_scan_batch (it then returns
 * the previous batch) keeps the walk of the newest batch overlapped; after fgpu_scan_end call it until -1.
 * FGPU_ERR_CAPACITY: *n_out = elements needed, nothing consumed. */
int fgpu_scan_take_stops(fgpu_ctx* ctx, fgpu_stop* out, uint64_t cap, uint64_t* n_out, int64_t* batch_seq);
/* Junction map after the scan, in CREATION order (inserting the records in this order into a
 * std::unordered_map<kmer_type,Junction> reproduces the reference's dump order,
 * utils/JunctionMap.cpp:588-593).  keys are the oriented k-mers (ReadKmer::getKmer). */
int fgpu_scan_junction_count(fgpu_ctx* ctx, uint64_t* n);
int fgpu_scan_download_junctions(fgpu_ctx* ctx, uint64_t* keys, fgpu_junction* recs, uint64_t cap, uint64_t* n_out);
/* Multi-GPU hand-over of the ordered state between consecutive read shards: export on rank r
 * (device buffer of n_entries * FGPU_TABLE_ENTRY_BYTES), import on rank r+1 before its first scan_batch. */
int fgpu_scan_table_entries(fgpu_ctx* ctx, uint64_t* n_entries);
/* The ORDER in which the reference dumps its junctions (JunctionMap::writeToFile, utils/JunctionMap.cpp:579-596: the iteration order of its
 * std::unordered_map, utils/JunctionMap.h:61), computed on the device for the keys of the last fgpu_scan_download_junctions -- which are in
 * creation order = the reference's insertion order.  The container's node list has a closed form (host/junction_order.h): between two rehashes
 * the nodes stand sorted by (first insertion into their bucket, own insertion), both latest first, and a rehash re-inserts the list as it
 * stands -- one radix sort per stretch, about 2 n elements in all.  The caller supplies WHEN its standard library rehashes and to how many
 * buckets (DumpOrder::schedule: asked of the library's own policy object): rehash_counts[j] nodes are present when the table goes to
 * rehash_buckets[j] buckets, first entry {0, buckets of the empty container}; the hash of a 64-bit key is the key (libstdc++).  order[i] =
 * index in creation order of the i-th junction dumped, for the first n keys (n <= the downloaded count; a prefix lets a host check the device
 * against its own container before trusting it).  Replaces a host-side replay that took 5.7 s for 2.95e7 junctions. */
int fgpu_scan_dump_order(fgpu_ctx* ctx, const uint64_t* rehash_counts, const uint64_t* rehash_buckets, uint64_t n_rehashes, uint64_t n, uint32_t* order);
int fgpu_scan_export_table(fgpu_ctx* ctx, void* dev_buf, uint64_t buf_bytes, uint64_t* n_entries);
int fgpu_scan_import_table(fgpu_ctx* ctx, const void* dev_buf, uint64_t n_entries, const fgpu_scan_stats* carried);
/* A PREVIEW of the table a later shard will be handed: an earlier state of the same ordered table (what the first shard has built
 * after part of its reads).  Imported before (or between) the calls of fgpu_scan_prepare -- before any walk --, it lets the pure stage evaluate the junction tests only where the walk
 * may stop, as it does in a streaming scan -- with an empty table every test of every position is evaluated (2.5x the probes).  Valid
 * because the table only grows along the file (keys are never removed, distances only rise), the same reason a batch may be previewed
 * against the table as of two batches earlier; the walk verifies the preview as always.  The hint is not part of the result: the
 * fgpu_scan_import_table that must follow replaces it, and the walk refuses to start while it is in place (FGPU_ERR_STATE). */
int fgpu_scan_import_hint(fgpu_ctx* ctx, const void* dev_buf, uint64_t n_entries);
/* A FRESHER preview (round 5): fgpu_scan_import_hint may be called again -- still before the real table --, e.g. with the table the shard below
 * has just been HANDED (one hop before this shard's own arrives), and fgpu_scan_refresh_prepared makes the in-map planes of every prepared batch
 * again against it, off the chain of walks.  When fgpu_scan_import_table then brings a later state of that very table, the walk only has to look
 * for the keys created since (found by their creation stamps, counted against the surplus of entries: a table that is NOT a later state of the
 * preview is noticed and handled the long way, exactly): 22 -> ~7 ms per 25 M reads on the chain.  FGPU_NO_DELTA_REFRESH=1: always the long way.
 * fgpu_diag_prepared_refresh (after fgpu_scan_end): [0] batches whose planes the walk made again in full, [1] batches it merged the new keys
 * into, [2] new keys of the last import, [3] with FGPU_DEBUG_DELTA_CHECK=1 (tests): plane words in which the merged planes differed from planes
 * made again in full (must be 0).  The filter of new keys also takes the keys this shard's own batches create as they are walked: the batches
 * behind them were prepared before any of the shard was walked. */
int fgpu_scan_refresh_prepared(fgpu_ctx* ctx);
int fgpu_diag_prepared_refresh(fgpu_ctx* ctx, uint64_t out[4]);
/* Round 6: fgpu_scan_refresh_prepared also makes, for every prepared batch, the plane of positions a link pass of the walk can find at all
 * (their hash is a candidate's by the planes as the preview left them); the merge of the new keys adds the positions that hit the filter of new
 * keys, and the walk links a window by visiting those positions only instead of probing the window table at every position (18 -> 3 ms per
 * 25 M reads on the chain; windows that follow another window of their batch are linked in full).  FGPU_NO_SPARSE_LINK=1: every window in full.
 * fgpu_diag_sparse_link (after fgpu_scan_end): [0] windows of prepared batches linked by the plane, [1] in full; FGPU_DEBUG_DELTA_CHECK=1 runs
 * the full pass behind every sparse one and counts differing lk words into fgpu_diag_prepared_refresh's [3]. */
int fgpu_diag_sparse_link(fgpu_ctx* ctx, uint64_t out[2]);
#define FGPU_TABLE_ENTRY_BYTES 32

/* The two pair filters as they stand on the device while a scan is open (after fgpu_scan_begin, which empties them): which = 0 the short
 * one (fgpu_scan_short_pairs), 1 the long one (fgpu_scan_long_pairs, FGPU_LONG_PAIRS_FILTER); tai / 8 bytes each.  For read shards walked one
 * after the other (several GPUs): the filters a shard ends with are what the next shard starts from -- the short filter only collects adds, the
 * long one is check-then-insert in file order, and a shard that begins at an even record finds no first end waiting.  The next shard copies
 * them in (fgpu_group_recv / fgpu_device_copy on the context's stream) before its first walk. */
int fgpu_scan_pairs_devptr(fgpu_ctx* ctx, int which, void** dptr, uint64_t* nbytes);

/* ---- device memory for the host's exchange buffers (HIP stays behind the ABI) -------------------- */
int fgpu_device_alloc(fgpu_ctx* ctx, uint64_t nbytes, void** dptr);      /* on the context's device */
int fgpu_device_free(fgpu_ctx* ctx, void* dptr);                          /* waits for the context's stream first */
int fgpu_device_copy(fgpu_ctx* ctx, void* dst_dev, const void* src_dev, uint64_t nbytes);   /* on the context's stream */
int fgpu_device_zero(fgpu_ctx* ctx, void* dst_dev, uint64_t nbytes);                        /* on the context's stream */

/* ---- several GPUs driven by ONE process: a group of contexts and their exchanges -----------------
 * The reference is a single process (src/Faucet.cpp:204-245); BASELINE.json's north_star shards its reads over the GPUs of one node.
 * A group ties N contexts together -- one per rank, each driven by its own host thread (the rule above), each on its own device, or
 * several on one device where there are fewer devices than ranks (tests) -- and moves DEVICE buffers between them on the contexts' own
 * streams: whatever a context has queued before an exchange is finished before its buffers are read, whatever it queues afterwards
 * sees what has arrived, and no call waits for a device.  Host threads do wait for each other (a receive for its send to be posted, a
 * send for its receive), which is the only synchronisation there is.
 *   FGPU_TRANSPORT_COPY  device-to-device copies inside the process (hipMemcpyPeerAsync over xGMI between devices), ordered by events
 *   FGPU_TRANSPORT_RCCL  ncclSend / ncclRecv on one communicator per rank (librccl.so is loaded when the first such group is made);
 *                        needs one device per rank (RCCL refuses two ranks on one device)
 * Every rank's thread makes the same sequence of collective calls; point-to-point calls pair up per (source, destination) in the order
 * they are made.  An error on one rank: fgpu_group_abort wakes every thread that waits and makes all further calls fail. */
typedef struct fgpu_group fgpu_group;
#define FGPU_TRANSPORT_COPY 0
#define FGPU_TRANSPORT_RCCL 1
int         fgpu_group_create(int n_ranks, int transport, fgpu_group** out);
void        fgpu_group_destroy(fgpu_group* g);                /* after every rank's thread has stopped using it */
/* rank's thread: ties ctx to the rank; returns when every rank has attached (RCCL: the communicators exist) */
int         fgpu_group_attach(fgpu_group* g, int rank, fgpu_ctx* ctx);
void        fgpu_group_abort(fgpu_group* g);
const char* fgpu_group_last_error(const fgpu_group* g, int rank);
int         fgpu_group_barrier(fgpu_group* g, int rank);       /* host threads only; no device is waited for */
/* bitmap := OR over ranks, in place on every rank (nbytes a multiple of 16, the same on all): reduce-scatter by slices -- rank q collects
 * slice q of everybody and reduces it with the OR kernel -- then all-gather of the reduced slices: 2 (N-1)/N of a bitmap crosses each
 * rank's links per call.  RCCL has no bitwise-OR reduction (rccl.h: ncclSum/Prod/Max/Min/Avg).  bloo2 after pass 1. */
int fgpu_group_or_allreduce(fgpu_group* g, int rank, void* bitmap_dev, uint64_t nbytes);
/* out := OR of the bitmaps of all ranks below `rank` (zero on rank 0); bitmap is left as it is.  The carried-in bloo1 of a read shard
 * (what the sequential run has in bloo1 when it reaches the shard). */
int fgpu_group_exclusive_prefix_or(fgpu_group* g, int rank, const void* bitmap_dev, void* out_dev, uint64_t nbytes);
/* point to point.  send: the buffer must stay as it is until the matching receive has run; the call returns once the receiver has
 * queued its copy (its thread has called fgpu_group_recv) and orders the sender's stream behind that copy.  send_async returns at once --
 * for a buffer nothing writes again before fgpu_group_flush (the table preview of DESIGN.md section 5).  probe: is a send from src waiting
 * (*nbytes its size)?  Never blocks. */
int fgpu_group_send(fgpu_group* g, int rank, int dst, const void* dev, uint64_t nbytes);
int fgpu_group_send_async(fgpu_group* g, int rank, int dst, const void* dev, uint64_t nbytes);
int fgpu_group_flush(fgpu_group* g, int rank);
int fgpu_group_recv(fgpu_group* g, int rank, int src, void* dev, uint64_t nbytes);
int fgpu_group_probe(fgpu_group* g, int rank, int src, int* waiting, uint64_t* nbytes);
/* the transport moves nbytes from a buffer of this rank to another buffer of this rank (RCCL: a send to itself and its receive in one group)
 * and the result is compared on the host: *ok = 1 iff every byte arrived.  What a box with one device can show of the RCCL transport. */
int fgpu_group_selftest(fgpu_group* g, int rank, uint64_t nbytes, int* ok);

/* ---- probes for tests (pure, no state change) ------------------------------------------------- */
/* For each of n k-mers (2-bit encoded, utils/Kmer.cpp:82-88,410-425): canonical form and
 * hA = oldHash(canon,0), hB = oldHash(canon,1) masked with tai-1 (utils/Bloom.h:134-145). */
int fgpu_probe_hash(fgpu_ctx* ctx, const uint64_t* kmers_host, uint64_t n, uint64_t* canon_out, uint64_t* hA_out, uint64_t* hB_out);
/* Bloom::oldContains (utils/Bloom.h:162-173) of n canonical k-mers against filter `which`. */
int fgpu_probe_contains(fgpu_ctx* ctx, int which, const uint64_t* canon_host, uint64_t n, uint8_t* out);

/* ---- Stage 3's Bloom probes, batched -----------------------------------------
 * The contig-graph stage walks the filter from every junction (JunctionMap::findNeighbor, utils/JunctionMap.cpp:231-412); its map
 * look-ups and its control flow stay on the host, its filter work is these pure functions of bloo2, here for n forward-strand
 * k-mers at once (one walk step of many junctions in lock-step):
 *   jcheck            JChecker::jcheck(kmer_type)              utils/JChecker.cpp:51-80      -> 0 / 1
 *   valid_extension   JunctionMap::getValidJExtension          utils/JunctionMap.cpp:474-490 -> -1 none, -2 several, else 0..3
 *   bloom_junction    JunctionMap::isBloomJunction             utils/JunctionMap.cpp:494-504 -> 0 / 1
 * (Bloom::oldContains itself is fgpu_probe_contains.) */
int fgpu_probe_jcheck(fgpu_ctx* ctx, const uint64_t* kmers_host, uint64_t n, int8_t* out);
int fgpu_probe_valid_extension(fgpu_ctx* ctx, const uint64_t* kmers_host, uint64_t n, int8_t* out);
int fgpu_probe_bloom_junction(fgpu_ctx* ctx, const uint64_t* kmers_host, uint64_t n, int8_t* out);

/* ---- Stage 3's walks, whole ------------------------------------------------------
 * JunctionMap::findNeighbor(Junction junc, kmer_type startKmer, int index) (utils/JunctionMap.cpp:231-412) for n (junction, extension)
 * pairs at once, one device lane per walk: every getValidJExtension, every DoubleKmer step and every junctionMap.find of the walk happens
 * on the device, against bloo2 as it stands (after a load, or fgpu_bloom_upload) and against the junction map handed over by
 * fgpu_stage3_set_junctions (k-mers as JunctionMap keys them, i.e. as fgpu_scan_download_junctions / a .junctions file give them; only the
 * five distances of a record are read, as in the reference).  The map stays on the device until it is set again.
 * Result per walk = the reference's return value: the BaseNode* it would build -- `node` 1: a junction was reached (`kmer` its key, `rindex`
 * the index it was entered through, 4 = backward), 0: a sink (`rindex` as the reference sets it) -- with the distance and the contig
 * length the reference reports through its pointer arguments.  `abort` 1 marks the calls in which the reference trips one of its asserts
 * (dist <= maxDist, validExtension >= 0) instead of returning; 2 = startKmer is not in the map or index is not 0..4.
 * n_probes (may be NULL): getValidJExtension evaluations made, each up to 4 x (Bloom::oldContains + JChecker::jcheck).
 * contigs_out (may be NULL): BfSearchResult::contig of every walk, the sequence getContig (utils/JunctionMap.cpp:133-227) appends to its
 * contig string: walk w's string is `len` bases in contigs_out[w * contig_stride_words ..], 2 bits per base in the reference's code
 * (A 0, C 1, T 2, G 3), 32 bases per word, the first base in the lowest bits; contig_stride_words >= fgpu_stage3_contig_words(k,
 * max_read_length). */
typedef struct {
    uint64_t kmer;
    int32_t  dist;
    int32_t  len;
    int8_t   node;
    int8_t   rindex;
    int8_t   abort;
    int8_t   reserved;
    int32_t  reserved2;
} fgpu_neighbor;
int fgpu_stage3_set_junctions(fgpu_ctx* ctx, const uint64_t* keys_host, const fgpu_junction* recs_host, uint64_t n);
int fgpu_stage3_find_neighbors(fgpu_ctx* ctx, const uint64_t* start_kmers_host, const int8_t* indices_host, uint64_t n,
                               int32_t max_read_length, fgpu_neighbor* out, uint64_t* n_probes, uint64_t* contigs_out,
                               uint64_t contig_stride_words);
uint64_t fgpu_stage3_contig_words(int32_t k, int32_t max_read_length);

/* ---- profiling --------------------------------------------------------------------------------- */
typedef struct {
    char     name[48];
    uint64_t launches;
    double   total_ms;       /* sum of HIP-event durations on the context's stream */
} fgpu_kernel_time;
/* Requires FGPU_FLAG_PROFILE.  Returns the number of distinct kernels; fills up to cap entries. */
int fgpu_kernel_times(fgpu_ctx* ctx, fgpu_kernel_time* out, int cap);
int fgpu_kernel_times_reset(fgpu_ctx* ctx);
/* Switch the event bracketing on or off between passes (what FGPU_FLAG_PROFILE sets at creation): the events cost 1.5 % of a step
 * (profiles/r04_profile_flag_ab.txt), so a caller times its steps without them and takes kernel times from separate, bracketed steps. */
int fgpu_profile_enable(fgpu_ctx* ctx, int on);

/* ---- measured ceilings (no reference counterpart) ---------------------------------
 * Streaming device-to-device copy of `bytes` (read + write counted), GB/s. */
int fgpu_diag_stream_copy(fgpu_ctx* ctx, uint64_t bytes, int iters, double* gb_per_s);
/* n_access independent random 32-bit accesses into a table of table_bytes (power of two) per iteration.
 * mode 0 = load, 1 = atomicMin (the load pass' first-set times), 2 = test-then-atomicOr (Bloom::add). */
/* NS1 as a whole probe CHAIN (round 3): n_items x Bloom::contains (n_hash dependent bit tests, early exit) directly, and with the first
 * level binned by filter slice and the survivors handed back as a dense list (faucet_amd/csrc/diag.hip).  Times per pass in ms; *equal = 1 iff
 * both forms give the same answers bit for bit (n_hash 2..8: with one hash function there is no chain).  Diagnostic: nothing of the path calls it. */
int fgpu_diag_binned_chain(fgpu_ctx* ctx, uint64_t table_bytes, uint64_t n_items, uint64_t slice_bytes, int n_hash, int fill_byte, int iters,
                           double* direct_ms, double* bin_ms, double* first_ms, double* rest_ms, double* survivors_share, int* equal);
/* Per-piece records of the key-ordered walk (measurement builds with -DFGPU_KO_TRACE; *n = 0 otherwise): 4 words per walked piece --
 * global piece number, start and end in 10 ns ticks, ticks waited for turns | lk positions << 48.  scripts/ko_trace.py reads them. */
int fgpu_diag_ko_trace(fgpu_ctx* ctx, uint64_t* out, uint64_t cap_records, uint64_t* n);
int fgpu_diag_ko_stamps(fgpu_ctx* ctx, uint64_t* out, uint64_t cap_words, uint64_t* n_pieces);   /* per-step stamps of one piece in 16 (-DFGPU_KO_TRACE) */
int fgpu_diag_random_access(fgpu_ctx* ctx, uint64_t table_bytes, uint64_t n_access, int mode, int iters, double* access_per_s);
/* The two access patterns of the marking kernel on the context's own tables of the last load pass (pair layout): random loads from the interleaved
 * filter words, random atomicMin into the first-set times; accesses per second; and both together as the kernel mixes them (three loads per item, three
 * atomics for six items in ten; items per second, may be NULL).  Between passes (the tables are rewritten by the next load). */
int fgpu_diag_load_tables(fgpu_ctx* ctx, uint64_t n_access, double* pair_loads_per_s, double* first_atomics_per_s, double* mixed_items_per_s);
/* The mixed pattern with k other allocations of the filter pair (held at once, then freed) against the context's first-set times: does the pair's place
 * decide the marking kernel's speed?  items_per_s[k].  Measurement (scripts/kinds_probe.py). */
int fgpu_diag_pair_placements(fgpu_ctx* ctx, int k, uint64_t n_items, double* items_per_s);
/* NS1's query-side blocking, measured: n_probes single-bit probes at pseudo-random positions of a table of table_bytes (power of two, <= 512 MiB),
 * once directly (one random load each, what the path's kernels do) and once binned by slice of slice_bytes (LDS-staged buckets per 4096 probes,
 * one queue per slice, slices probed by the workgroups of the XCD whose L2 then holds them).  Rates in probes/s; the binned one covers both of
 * its kernels, whose mean times come back separately. */
int fgpu_diag_binned_probes(fgpu_ctx* ctx, uint64_t table_bytes, uint64_t n_probes, uint64_t slice_bytes, int iters, double* direct_per_s,
                            double* binned_per_s, double* bin_ms, double* probe_ms);
/* How often the library has scanned a pass's batches again by itself (lazy junction tests, see fgpu_scan_batch) since the context was made. */
/* after fgpu_scan_end: pieces of large clusters probed for the out-of-order walk, by outcome: [0] order-free, [1] would create a junction,
 * [2] would raise a distance, [3] crosses positions whose junction tests the preview left out */
int fgpu_diag_walk_probe(fgpu_ctx* ctx, uint64_t out[4]);
int fgpu_diag_scan_replays(fgpu_ctx* ctx, uint64_t* replays);
/* How often the host thread has waited for the device (hipStreamSynchronize / hipEventSynchronize inside the library) since the current or last pass
 * began (fgpu_load_begin / fgpu_scan_begin), and for how long in all (ms, may be NULL).  Every wait is also a chance for a busy host to schedule the
 * thread late: what a pass costs beside other tenants of the box grows with this count, not with the device's work. */
int fgpu_diag_host_waits(fgpu_ctx* ctx, uint64_t* waits, double* ms);
/* after fgpu_scan_end: [0] junction tests the walk had to run itself (the preview had left the position out) that came out TRUE at a
 * k-mer no piece of the window had registered -- the walk goes on and the window is checked afterwards --, [1] of those, the ones whose
 * k-mer was then found on another piece of the same window: only these void a lazy scan (the library scans its journal again, see above),
 * [2] noted positions the check itself passed over (= [0] when it works: a self-test of the sweep). */
int fgpu_diag_late_flags(fgpu_ctx* ctx, uint64_t out[3]);
/* after a scan with the long pair filter on the device: [0] items (first-end k-mers of read pairs with two non-empty lists), [1] of those, found
 * paired against the filter as their batch found it, [2] addPair calls, [3] evaluation rounds over all batches, [4] most rounds one batch
 * needed, [5] batches */
int fgpu_diag_long_pairs(fgpu_ctx* ctx, uint64_t out[6]);
/* after fgpu_scan_end: the optimistic walk of large clusters: [0] pieces it walked, [1] rounds it ran, [2] windows it settled,
 * [3] windows it left to the key-ordered walk (rounds that did not settle, full tables), [4] piece-rounds in which a piece kept its log (no
 * earlier piece had changed what it reads), [5] windows whose large clusters outgrew the tables of both walks (walked by cluster) */
int fgpu_diag_ovw(fgpu_ctx* ctx, uint64_t out[6]);
/* ... and how full its event tables got: *high_water = the most entries a round of any window of the scan held, *capacity = entries per table as
 * they stand.  The library grows the tables (x 2 up to 2^27 entries, 32 bytes each) once a round has filled a quarter, so that coverage of many
 * hundred-fold inside repeats does not first overflow them (exact either way: an overflowing window goes to the key-ordered walk, slowly).
 * FGPU_OVW_EV_LOG2 = log2 of the entries a context starts with (default 23). */
int fgpu_diag_ovw_tables(fgpu_ctx* ctx, uint64_t* high_water, uint64_t* capacity);
/* Where the last load pass settled its occurrences (measurement: which kernel performs the reference's bloo2 sets): *in_mark = occurrences
 * whose bits were all in the carried-in state and that the marking kernel itself routed to bloo2, *pending = occurrences left to the
 * first-set-time resolution.  Valid after fgpu_load_end, until the next pass begins. */
int fgpu_diag_load_split(fgpu_ctx* ctx, uint64_t* in_mark, uint64_t* pending);
/* Memory clock (kHz), bus width (bits), L2 bytes and CU count as the HIP runtime reports them. */
int fgpu_diag_device_attr(fgpu_ctx* ctx, int32_t* mem_clock_khz, int32_t* mem_bus_bits, int32_t* l2_bytes, int32_t* compute_units);

#ifdef __cplusplus
}
#endif
#endif /* FAUCET_GPU_H */
