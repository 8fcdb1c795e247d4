// group.hip — several contexts of ONE process (one per GPU, each with its own host thread) and the exchanges between them.
//
// The reference is a single process on one core (src/Faucet.cpp:204-245); BASELINE.json's north_star shards its reads over the GPUs of a
// node: contiguous file-order shards, an exclusive prefix-OR of the shards' bloo1 and an OR-allreduce of bloo2 in pass 1 (SURVEY.md A.5:
// utils/Bloom.cpp:289-299 made exact over shards), the ordered junction walk handed from shard to shard in pass 2 (src/ReadScanner.cpp:61-231).
// faucet_amd/sharded.py does that with one PROCESS per GPU over torch.distributed; this file is what a C++ host (the `faucet` command line,
// integration/faucet_binding.cpp) uses instead: the same slice schedule, moved by
//   * device-to-device copies between the contexts of the process (hipMemcpyPeerAsync: xGMI between devices, a plain copy on one), or
//   * RCCL called directly (ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on one communicator per rank), librccl loaded at run time.
// Everything is queued on the contexts' own streams: an exchange follows the kernels that make its input and precedes the kernels that read
// its output without any wait for a device.  Host threads rendezvous through per-(source, destination) mailboxes.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <deque>
#include <mutex>

#include "fgpu_ctx.h"

namespace {

struct Msg {
    const void* ptr = nullptr;
    uint64_t nbytes = 0;
    int device = 0;
    hipEvent_t ready = nullptr;   // sender's stream: the buffer holds what is sent
    hipEvent_t done = nullptr;    // receiver's stream: the copy out of the buffer has run
    bool copied = false;          // `done` is recorded (the receiver's thread has queued its copy)
    bool wire = false;            // RCCL: the bytes travel by ncclSend, the mailbox only announces them
};

struct Channel {
    std::deque<Msg*> q;           // posted, not yet received
};

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

struct RankState {
    fgpu_ctx* ctx = nullptr;
    int device = -1;
    hipStream_t xstream = nullptr;        // side stream of asynchronous sends under RCCL (the context's stream goes on meanwhile)
    std::vector<hipEvent_t> events;       // every event this rank made for the group (destroyed with it)
    std::vector<Msg*> outstanding;        // asynchronous sends not yet flushed
    void* scratch[2] = {nullptr, nullptr};
    uint64_t scratch_bytes[2] = {0, 0};
    ncclComm_t comm = nullptr;
    std::string err;
};

}  // namespace

struct fgpu_group {
    int n = 0, transport = FGPU_TRANSPORT_COPY;
    std::mutex m;
    std::condition_variable cv;
    std::vector<RankState> ranks;
    std::vector<Channel> chan;            // [src * n + dst]
    int attached = 0;
    bool ready = false, aborted = false;
    int bar_count = 0;
    uint64_t bar_gen = 0;
    RcclApi rccl;
    std::string err;                      // failures that belong to no rank (creation, the communicators)
};

namespace {

int gfail(fgpu_group* g, int rank, int rc, const std::string& what) {
    std::lock_guard<std::mutex> lk(g->m);
    if (rank >= 0 && rank < g->n) g->ranks[rank].err = what; else g->err = what;
    return rc;
}

#define GHIP(call)                                                                                             \
    do {                                                                                                       \
        hipError_t e__ = (call);                                                                               \
        if (e__ != hipSuccess) {                                                                               \
            char b__[512];                                                                                     \
            snprintf(b__, sizeof(b__), "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
            return gfail(g, rank, FGPU_ERR_HIP, b__);                                                          \
        }                                                                                                      \
    } while (0)

#define GNCCL(call)                                                                                            \
    do {                                                                                                       \
        ncclResult_t r__ = (call);                                                                             \
        if (r__ != ncclSuccess) {                                                                              \
            char b__[512];                                                                                     \
            snprintf(b__, sizeof(b__), "%s failed: %s (%s:%d)", #call, g->rccl.GetErrorString ? g->rccl.GetErrorString(r__) : "?", __FILE__, __LINE__); \
            return gfail(g, rank, FGPU_ERR_HIP, b__);                                                          \
        }                                                                                                      \
    } while (0)

bool bad_rank(const fgpu_group* g, int rank) { return !g || rank < 0 || rank >= g->n; }

int check_rank(fgpu_group* g, int rank) {
    if (bad_rank(g, rank)) return FGPU_ERR_ARG;
    std::lock_guard<std::mutex> lk(g->m);
    if (g->aborted) { g->ranks[rank].err = "the group was aborted (another rank failed)"; return FGPU_ERR_STATE; }
    if (!g->ready || !g->ranks[rank].ctx) { g->ranks[rank].err = "rank not attached (fgpu_group_attach)"; return FGPU_ERR_STATE; }
    return FGPU_OK;
}

int new_event(fgpu_group* g, int rank, hipEvent_t* ev) {
    GHIP(hipEventCreateWithFlags(ev, hipEventDisableTiming));
    g->ranks[rank].events.push_back(*ev);
    return FGPU_OK;
}

int load_rccl(fgpu_group* g) {
    RcclApi& a = g->rccl;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        a.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (a.lib) break;
    }
    if (!a.lib) { g->err = std::string("RCCL transport: librccl.so could not be loaded: ") + (dlerror() ? dlerror() : "?"); return FGPU_ERR_HIP; }
    bool ok = true;
    auto sym = [&](const char* n) { void* p = dlsym(a.lib, n); if (!p) ok = false; return p; };
    a.CommInitAll = (decltype(a.CommInitAll))sym("ncclCommInitAll");
    a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
    a.GroupStart = (decltype(a.GroupStart))sym("ncclGroupStart");
    a.GroupEnd = (decltype(a.GroupEnd))sym("ncclGroupEnd");
    a.Send = (decltype(a.Send))sym("ncclSend");
    a.Recv = (decltype(a.Recv))sym("ncclRecv");
    a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
    if (!ok) { g->err = "RCCL transport: librccl.so lacks a symbol this library calls (ncclCommInitAll, ncclSend, ncclRecv, ncclGroupStart/End)"; return FGPU_ERR_HIP; }
    return FGPU_OK;
}

struct Xfer { int peer; void* ptr; uint64_t nbytes; };

// [lo, hi) byte ranges of the n slices of a bitmap: equal, 16-byte aligned (the OR kernel's granule), the last ones short or empty
void slices(uint64_t nbytes, int n, std::vector<uint64_t>& lo, std::vector<uint64_t>& hi) {
    uint64_t step = (nbytes + (uint64_t)n - 1) / (uint64_t)n;
    step = (step + 15) & ~15ULL;
    lo.resize((size_t)n);
    hi.resize((size_t)n);
    for (int q = 0; q < n; q++) {
        lo[(size_t)q] = std::min<uint64_t>((uint64_t)q * step, nbytes);
        hi[(size_t)q] = std::min<uint64_t>((uint64_t)(q + 1) * step, nbytes);
    }
}

int scratch(fgpu_group* g, int rank, int which, uint64_t bytes, void** out) {
    RankState& r = g->ranks[rank];
    if (r.scratch_bytes[which] < bytes) {
        if (r.scratch[which]) {
            GHIP(hipStreamSynchronize(r.ctx->stream));
            GHIP(hipFree(r.scratch[which]));
            r.scratch[which] = nullptr;
            r.scratch_bytes[which] = 0;
        }
        if (hipMalloc(&r.scratch[which], bytes) != hipSuccess) {
            r.scratch[which] = nullptr;
            return gfail(g, rank, FGPU_ERR_NOMEM, "hipMalloc of an exchange buffer failed");
        }
        r.scratch_bytes[which] = bytes;
    }
    *out = r.scratch[which];
    return FGPU_OK;
}

// a send is announced in the mailbox of (rank -> dst); under the copy transport the message IS the transfer (the receiver copies out of ptr)
int post(fgpu_group* g, int rank, int dst, const void* ptr, uint64_t nbytes, hipEvent_t ready, bool wire, Msg** out) {
    Msg* msg = new Msg();
    msg->ptr = ptr;
    msg->nbytes = nbytes;
    msg->device = g->ranks[rank].device;
    msg->ready = ready;
    msg->wire = wire;
    {
        std::lock_guard<std::mutex> lk(g->m);
        g->chan[(size_t)rank * g->n + dst].q.push_back(msg);
    }
    g->cv.notify_all();
    *out = msg;
    return FGPU_OK;
}

// the next message of (src -> rank), waited for
int take(fgpu_group* g, int rank, int src, uint64_t nbytes, Msg** out) {
    std::unique_lock<std::mutex> lk(g->m);
    Channel& c = g->chan[(size_t)src * g->n + rank];
    g->cv.wait(lk, [&] { return g->aborted || !c.q.empty(); });
    if (g->aborted) { g->ranks[rank].err = "the group was aborted while a receive waited for its send"; return FGPU_ERR_STATE; }
    Msg* msg = c.q.front();
    if (msg->nbytes != nbytes) {
        char b[160];
        snprintf(b, sizeof(b), "receive of %llu bytes from rank %d meets a send of %llu bytes", (unsigned long long)nbytes, src, (unsigned long long)msg->nbytes);
        g->ranks[rank].err = b;
        return FGPU_ERR_ARG;
    }
    c.q.pop_front();
    *out = msg;
    return FGPU_OK;
}

// receiver side of the copy transport: the copy runs on this rank's stream behind the sender's `ready`; `done` tells the sender
int copy_in(fgpu_group* g, int rank, Msg* msg, void* dst) {
    RankState& r = g->ranks[rank];
    hipStream_t st = r.ctx->stream;
    GHIP(hipStreamWaitEvent(st, msg->ready, 0));
    if (msg->nbytes) {
        if (msg->device == r.device) GHIP(hipMemcpyAsync(dst, msg->ptr, msg->nbytes, hipMemcpyDeviceToDevice, st));
        else GHIP(hipMemcpyPeerAsync(dst, r.device, msg->ptr, msg->device, msg->nbytes, st));
    }
    hipEvent_t done;
    if (int rc = new_event(g, rank, &done)) return rc;
    GHIP(hipEventRecord(done, st));
    {
        std::lock_guard<std::mutex> lk(g->m);
        msg->done = done;
        msg->copied = true;
    }
    g->cv.notify_all();
    return FGPU_OK;
}

// sender side: wait (host) until the receiver has queued its copy, then order this rank's stream behind it -- the buffer may be rewritten
int settle(fgpu_group* g, int rank, Msg* msg) {
    {
        std::unique_lock<std::mutex> lk(g->m);
        g->cv.wait(lk, [&] { return g->aborted || msg->copied; });
        if (!msg->copied) { g->ranks[rank].err = "the group was aborted while a send waited for its receive"; return FGPU_ERR_STATE; }
    }
    if (msg->done) GHIP(hipStreamWaitEvent(g->ranks[rank].ctx->stream, msg->done, 0));
    delete msg;
    return FGPU_OK;
}

// One grouped exchange of this rank: every send and receive of the list, matched with the other ranks' lists per (source, destination) in order.
// Copy transport: post all sends, run all receives, settle all sends -- no rank waits for another before its own sends are posted, so
// lists that match cannot deadlock.  RCCL: the list is one ncclGroup on the context's stream.
int exchange(fgpu_group* g, int rank, const std::vector<Xfer>& sends, const std::vector<Xfer>& recvs) {
    RankState& r = g->ranks[rank];
    GHIP(hipSetDevice(r.device));
    if (g->transport == FGPU_TRANSPORT_RCCL) {
        if (sends.empty() && recvs.empty()) return FGPU_OK;
        GNCCL(g->rccl.GroupStart());
        for (const Xfer& s : sends) GNCCL(g->rccl.Send(s.ptr, (size_t)s.nbytes, ncclUint8, s.peer, r.comm, r.ctx->stream));
        for (const Xfer& v : recvs) GNCCL(g->rccl.Recv(v.ptr, (size_t)v.nbytes, ncclUint8, v.peer, r.comm, r.ctx->stream));
        GNCCL(g->rccl.GroupEnd());
        return FGPU_OK;
    }
    std::vector<Msg*> posted;
    int rc = FGPU_OK;
    if (!sends.empty()) {
        hipEvent_t ready;
        if ((rc = new_event(g, rank, &ready))) return rc;
        GHIP(hipEventRecord(ready, r.ctx->stream));
        for (const Xfer& s : sends) {
            Msg* msg = nullptr;
            if ((rc = post(g, rank, s.peer, s.ptr, s.nbytes, ready, false, &msg))) return rc;
            posted.push_back(msg);
        }
    }
    for (const Xfer& v : recvs) {
        Msg* msg = nullptr;
        if ((rc = take(g, rank, v.peer, v.nbytes, &msg))) return rc;
        if ((rc = copy_in(g, rank, msg, v.ptr))) return rc;
    }
    for (Msg* msg : posted)
        if ((rc = settle(g, rank, msg))) return rc;
    return FGPU_OK;
}

}  // namespace

extern "C" {

int fgpu_group_create(int n_ranks, int transport, fgpu_group** out) {
    if (!out || n_ranks < 1 || n_ranks > 64 || (transport != FGPU_TRANSPORT_COPY && transport != FGPU_TRANSPORT_RCCL)) return FGPU_ERR_ARG;
    fgpu_group* g = new fgpu_group();
    g->n = n_ranks;
    g->transport = transport;
    g->ranks.resize((size_t)n_ranks);
    g->chan.resize((size_t)n_ranks * n_ranks);
    *out = g;
    if (transport == FGPU_TRANSPORT_RCCL)
        if (int rc = load_rccl(g)) return rc;     // (the group exists so that fgpu_group_last_error(g, -1) can say why; the caller destroys it)
    return FGPU_OK;
}

void fgpu_group_destroy(fgpu_group* g) {
    if (!g) return;
    std::vector<Msg*> left;                  // messages nobody settled (an aborted run): each is in a mailbox, on a sender's list, or both
    auto note = [&left](Msg* m) { for (Msg* o : left) if (o == m) return; left.push_back(m); };
    for (RankState& r : g->ranks) {
        if (r.device >= 0) (void)hipSetDevice(r.device);
        if (r.ctx && r.ctx->stream) (void)hipStreamSynchronize(r.ctx->stream);
        if (r.xstream) { (void)hipStreamSynchronize(r.xstream); (void)hipStreamDestroy(r.xstream); }
        if (r.comm && g->rccl.CommDestroy) (void)g->rccl.CommDestroy(r.comm);
        for (hipEvent_t e : r.events) (void)hipEventDestroy(e);
        for (void* p : r.scratch) if (p) (void)hipFree(p);
        for (Msg* msg : r.outstanding) note(msg);
    }
    for (Channel& c : g->chan)
        for (Msg* msg : c.q) note(msg);
    for (Msg* msg : left) delete msg;
    // (librccl stays loaded: unloading a library that owns device state at this point gains nothing)
    delete g;
}

const char* fgpu_group_last_error(const fgpu_group* g, int rank) {
    if (!g) return "no group";
    if (rank >= 0 && rank < g->n && !g->ranks[(size_t)rank].err.empty()) return g->ranks[(size_t)rank].err.c_str();
    return g->err.c_str();
}

void fgpu_group_abort(fgpu_group* g) {
    if (!g) return;
    {
        std::lock_guard<std::mutex> lk(g->m);
        g->aborted = true;
    }
    g->cv.notify_all();
}

int fgpu_group_attach(fgpu_group* g, int rank, fgpu_ctx* ctx) {
    if (bad_rank(g, rank) || !ctx) return FGPU_ERR_ARG;
    RankState& r = g->ranks[(size_t)rank];
    GHIP(hipSetDevice(ctx->prm.device));
    {
        std::unique_lock<std::mutex> lk(g->m);
        if (r.ctx) { r.err = "rank attached twice"; return FGPU_ERR_STATE; }
        r.ctx = ctx;
        r.device = ctx->prm.device;
        g->attached++;
        if (g->attached == g->n) {
            if (g->transport == FGPU_TRANSPORT_RCCL) {     // the last rank to arrive makes all communicators (ncclCommInitAll: one call, one process)
                std::vector<int> devs;
                std::vector<ncclComm_t> comms((size_t)g->n);
                for (RankState& q : g->ranks) devs.push_back(q.device);
                bool distinct = true;
                for (size_t a = 0; a < devs.size(); a++)
                    for (size_t b = a + 1; b < devs.size(); b++) if (devs[a] == devs[b]) distinct = false;
                ncclResult_t res = distinct ? g->rccl.CommInitAll(comms.data(), g->n, devs.data()) : ncclInvalidUsage;
                if (res != ncclSuccess) {
                    g->err = std::string("RCCL transport: ") + (distinct ? std::string("ncclCommInitAll failed: ") + g->rccl.GetErrorString(res)
                                                                         : std::string("two ranks share a device (RCCL needs one device per rank; use the copy transport)"));
                    g->aborted = true;
                } else {
                    for (int q = 0; q < g->n; q++) g->ranks[(size_t)q].comm = comms[(size_t)q];
                }
                (void)hipSetDevice(ctx->prm.device);
            }
            g->ready = true;
            g->cv.notify_all();
        } else {
            g->cv.wait(lk, [&] { return g->ready || g->aborted; });
        }
        if (g->aborted) { if (r.err.empty()) r.err = g->err.empty() ? "the group was aborted while ranks attached" : g->err; return FGPU_ERR_STATE; }
    }
    if (g->transport == FGPU_TRANSPORT_COPY) {
        // direct xGMI copies between the devices of the group (a copy between devices without peer access is staged by the runtime)
        for (int q = 0; q < g->n; q++) {
            const int other = g->ranks[(size_t)q].device;
            int can = 0;
            if (other != r.device && hipDeviceCanAccessPeer(&can, r.device, other) == hipSuccess && can) {
                hipError_t e = hipDeviceEnablePeerAccess(other, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
            }
        }
        (void)hipGetLastError();
    } else {
        GHIP(hipStreamCreateWithFlags(&r.xstream, hipStreamNonBlocking));
    }
    return FGPU_OK;
}

int fgpu_group_barrier(fgpu_group* g, int rank) {
    if (bad_rank(g, rank)) return FGPU_ERR_ARG;
    std::unique_lock<std::mutex> lk(g->m);
    if (g->aborted) { g->ranks[(size_t)rank].err = "the group was aborted"; return FGPU_ERR_STATE; }
    const uint64_t gen = g->bar_gen;
    if (++g->bar_count == g->n) {
        g->bar_count = 0;
        g->bar_gen++;
        g->cv.notify_all();
        return FGPU_OK;
    }
    g->cv.wait(lk, [&] { return g->aborted || g->bar_gen != gen; });
    if (g->bar_gen == gen) { g->ranks[(size_t)rank].err = "the group was aborted at a barrier"; return FGPU_ERR_STATE; }
    return FGPU_OK;
}

int fgpu_group_or_allreduce(fgpu_group* g, int rank, void* bitmap_dev, uint64_t nbytes) {
    if (int rc = check_rank(g, rank)) return rc;
    if (!bitmap_dev || (nbytes & 15)) return gfail(g, rank, FGPU_ERR_ARG, "or_allreduce: a device bitmap of a multiple of 16 bytes");
    if (g->n == 1) return FGPU_OK;
    RankState& r = g->ranks[(size_t)rank];
    GHIP(hipSetDevice(r.device));
    std::vector<uint64_t> lo, hi;
    slices(nbytes, g->n, lo, hi);
    const uint64_t mine = hi[(size_t)rank] - lo[(size_t)rank];
    char* bm = (char*)bitmap_dev;
    void* stage_v = nullptr;
    if (int rc = scratch(g, rank, 0, std::max<uint64_t>(mine, 16) * (uint64_t)(g->n - 1), &stage_v)) return rc;
    char* stage = (char*)stage_v;
    // reduce-scatter: slice q of this rank's bitmap goes to rank q, slice `rank` of everybody else's comes here ...
    std::vector<Xfer> sends, recvs;
    int i = 0;
    for (int q = 0; q < g->n; q++) {
        if (q == rank) continue;
        if (hi[(size_t)q] > lo[(size_t)q]) sends.push_back(Xfer{q, bm + lo[(size_t)q], hi[(size_t)q] - lo[(size_t)q]});
        if (mine) recvs.push_back(Xfer{q, stage + (uint64_t)i * mine, mine});
        i++;
    }
    if (int rc = exchange(g, rank, sends, recvs)) return rc;
    // ... and is reduced with the OR kernel on the context's stream (behind the copies that have just been queued there)
    for (int s = 0; mine && s < g->n - 1; s++) {
        if (int rc = fgpu_util_or(r.ctx, bm + lo[(size_t)rank], stage + (uint64_t)s * mine, mine)) return gfail(g, rank, rc, r.ctx->err);
    }
    // all-gather: the reduced slice to everybody, theirs straight into place
    sends.clear();
    recvs.clear();
    for (int q = 0; q < g->n; q++) {
        if (q == rank) continue;
        if (mine) sends.push_back(Xfer{q, bm + lo[(size_t)rank], mine});
        if (hi[(size_t)q] > lo[(size_t)q]) recvs.push_back(Xfer{q, bm + lo[(size_t)q], hi[(size_t)q] - lo[(size_t)q]});
    }
    return exchange(g, rank, sends, recvs);
}

int fgpu_group_exclusive_prefix_or(fgpu_group* g, int rank, const void* bitmap_dev, void* out_dev, uint64_t nbytes) {
    if (int rc = check_rank(g, rank)) return rc;
    if (!bitmap_dev || !out_dev || (nbytes & 15)) return gfail(g, rank, FGPU_ERR_ARG, "exclusive_prefix_or: device bitmaps of a multiple of 16 bytes");
    RankState& r = g->ranks[(size_t)rank];
    GHIP(hipSetDevice(r.device));
    hipStream_t st = r.ctx->stream;
    if (g->n == 1) {
        GHIP(hipMemsetAsync(out_dev, 0, nbytes, st));
        return FGPU_OK;
    }
    std::vector<uint64_t> lo, hi;
    slices(nbytes, g->n, lo, hi);
    const uint64_t mine = hi[(size_t)rank] - lo[(size_t)rank], cell = std::max<uint64_t>(mine, 16);
    const char* bm = (const char*)bitmap_dev;
    char* out = (char*)out_dev;
    void *stage_v = nullptr, *pref_v = nullptr;
    if (int rc = scratch(g, rank, 0, cell * (uint64_t)g->n, &stage_v)) return rc;     // slice `rank` of every rank, in rank order
    if (int rc = scratch(g, rank, 1, cell * (uint64_t)g->n, &pref_v)) return rc;      // what goes back to rank q: OR over ranks < q of that slice
    char *stage = (char*)stage_v, *pref = (char*)pref_v;
    if (mine) GHIP(hipMemcpyAsync(stage + (uint64_t)rank * mine, bm + lo[(size_t)rank], mine, hipMemcpyDeviceToDevice, st));
    std::vector<Xfer> sends, recvs;
    for (int q = 0; q < g->n; q++) {
        if (q == rank) continue;
        if (hi[(size_t)q] > lo[(size_t)q]) sends.push_back(Xfer{q, (void*)(bm + lo[(size_t)q]), hi[(size_t)q] - lo[(size_t)q]});
        if (mine) recvs.push_back(Xfer{q, stage + (uint64_t)q * mine, mine});
    }
    if (int rc = exchange(g, rank, sends, recvs)) return rc;
    if (mine) {
        GHIP(hipMemsetAsync(pref, 0, mine, st));
        for (int q = 1; q < g->n; q++) {                       // running OR: pref[q] = pref[q-1] | stage[q-1]
            GHIP(hipMemcpyAsync(pref + (uint64_t)q * mine, pref + (uint64_t)(q - 1) * mine, mine, hipMemcpyDeviceToDevice, st));
            if (int rc = fgpu_util_or(r.ctx, pref + (uint64_t)q * mine, stage + (uint64_t)(q - 1) * mine, mine)) return gfail(g, rank, rc, r.ctx->err);
        }
    }
    // rank 0's prefix is empty: nothing is sent to it, it zero-fills
    sends.clear();
    recvs.clear();
    for (int q = 0; q < g->n; q++) {
        if (q == rank) continue;
        if (mine && q != 0) sends.push_back(Xfer{q, pref + (uint64_t)q * mine, mine});
        if (rank != 0 && hi[(size_t)q] > lo[(size_t)q]) recvs.push_back(Xfer{q, out + lo[(size_t)q], hi[(size_t)q] - lo[(size_t)q]});
    }
    if (int rc = exchange(g, rank, sends, recvs)) return rc;
    if (rank == 0) GHIP(hipMemsetAsync(out, 0, nbytes, st));
    else if (mine) GHIP(hipMemcpyAsync(out + lo[(size_t)rank], pref + (uint64_t)rank * mine, mine, hipMemcpyDeviceToDevice, st));
    return FGPU_OK;
}

int fgpu_group_send(fgpu_group* g, int rank, int dst, const void* dev, uint64_t nbytes) {
    if (int rc = check_rank(g, rank)) return rc;
    if (bad_rank(g, dst) || dst == rank || (nbytes && !dev)) return gfail(g, rank, FGPU_ERR_ARG, "send: another rank of the group and a device buffer");
    RankState& r = g->ranks[(size_t)rank];
    GHIP(hipSetDevice(r.device));
    if (g->transport == FGPU_TRANSPORT_RCCL) {
        // announced in the mailbox (fgpu_group_probe sees it, the sizes are checked), moved by ncclSend on the context's stream
        Msg* msg = nullptr;
        if (int rc = post(g, rank, dst, dev, nbytes, nullptr, true, &msg)) return rc;
        if (nbytes) GNCCL(g->rccl.Send(dev, (size_t)nbytes, ncclUint8, dst, r.comm, r.ctx->stream));
        return FGPU_OK;      // (the receiver frees the announcement; stream order keeps the buffer valid: later work of this rank follows the send)
    }
    return exchange(g, rank, std::vector<Xfer>{Xfer{dst, (void*)dev, nbytes}}, std::vector<Xfer>());
}

int fgpu_group_send_async(fgpu_group* g, int rank, int dst, const void* dev, uint64_t nbytes) {
    if (int rc = check_rank(g, rank)) return rc;
    if (bad_rank(g, dst) || dst == rank || (nbytes && !dev)) return gfail(g, rank, FGPU_ERR_ARG, "send_async: another rank of the group and a device buffer");
    RankState& r = g->ranks[(size_t)rank];
    GHIP(hipSetDevice(r.device));
    hipEvent_t ready;
    if (int rc = new_event(g, rank, &ready)) return rc;
    GHIP(hipEventRecord(ready, r.ctx->stream));
    Msg* msg = nullptr;
    if (g->transport == FGPU_TRANSPORT_RCCL) {
        // on the side stream, behind what the context's stream holds now: the context goes on while the receiver has not asked yet
        if (int rc = post(g, rank, dst, dev, nbytes, nullptr, true, &msg)) return rc;
        GHIP(hipStreamWaitEvent(r.xstream, ready, 0));
        if (nbytes) GNCCL(g->rccl.Send(dev, (size_t)nbytes, ncclUint8, dst, r.comm, r.xstream));
        return FGPU_OK;
    }
    if (int rc = post(g, rank, dst, dev, nbytes, ready, false, &msg)) return rc;
    r.outstanding.push_back(msg);
    return FGPU_OK;
}

int fgpu_group_flush(fgpu_group* g, int rank) {
    if (bad_rank(g, rank)) return FGPU_ERR_ARG;
    RankState& r = g->ranks[(size_t)rank];
    if (!r.ctx) return FGPU_OK;
    GHIP(hipSetDevice(r.device));
    int rc = FGPU_OK;
    std::vector<Msg*> out;
    out.swap(r.outstanding);
    for (size_t i = 0; i < out.size(); i++) {
        if (!rc) rc = settle(g, rank, out[i]);
        else {                                       // aborted: a message still queued is taken back, one a receiver holds is left to it
            std::lock_guard<std::mutex> lk(g->m);
            for (Channel& c : g->chan)
                for (size_t k = 0; k < c.q.size(); k++)
                    if (c.q[k] == out[i]) { c.q.erase(c.q.begin() + (ptrdiff_t)k); delete out[i]; break; }
        }
    }
    if (r.xstream) {                                 // RCCL: later work of the context follows the side stream's sends
        hipEvent_t ev;
        if (int rc2 = new_event(g, rank, &ev)) return rc2;
        GHIP(hipEventRecord(ev, r.xstream));
        GHIP(hipStreamWaitEvent(r.ctx->stream, ev, 0));
    }
    return rc;
}

int fgpu_group_recv(fgpu_group* g, int rank, int src, void* dev, uint64_t nbytes) {
    if (int rc = check_rank(g, rank)) return rc;
    if (bad_rank(g, src) || src == rank || (nbytes && !dev)) return gfail(g, rank, FGPU_ERR_ARG, "recv: another rank of the group and a device buffer");
    RankState& r = g->ranks[(size_t)rank];
    GHIP(hipSetDevice(r.device));
    Msg* msg = nullptr;
    if (int rc = take(g, rank, src, nbytes, &msg)) return rc;
    if (msg->wire) {
        delete msg;
        if (nbytes) GNCCL(g->rccl.Recv(dev, (size_t)nbytes, ncclUint8, src, r.comm, r.ctx->stream));
        return FGPU_OK;
    }
    return copy_in(g, rank, msg, dev);
}

int fgpu_group_probe(fgpu_group* g, int rank, int src, int* waiting, uint64_t* nbytes) {
    if (bad_rank(g, rank) || bad_rank(g, src) || !waiting) return FGPU_ERR_ARG;
    std::lock_guard<std::mutex> lk(g->m);
    if (g->aborted) { g->ranks[(size_t)rank].err = "the group was aborted"; return FGPU_ERR_STATE; }
    Channel& c = g->chan[(size_t)src * g->n + rank];
    *waiting = c.q.empty() ? 0 : 1;
    if (nbytes) *nbytes = c.q.empty() ? 0 : c.q.front()->nbytes;
    return FGPU_OK;
}

int fgpu_group_selftest(fgpu_group* g, int rank, uint64_t nbytes, int* ok) {
    if (int rc = check_rank(g, rank)) return rc;
    if (!ok || !nbytes) return FGPU_ERR_ARG;
    *ok = 0;
    RankState& r = g->ranks[(size_t)rank];
    GHIP(hipSetDevice(r.device));
    hipStream_t st = r.ctx->stream;
    void* buf = nullptr;
    if (int rc = scratch(g, rank, 0, 2 * nbytes, &buf)) return rc;
    std::vector<uint8_t> src((size_t)nbytes), back((size_t)nbytes);
    for (uint64_t i = 0; i < nbytes; i++) src[(size_t)i] = (uint8_t)((i * 2654435761ULL >> 7) ^ (uint64_t)rank);
    GHIP(hipMemcpyAsync(buf, src.data(), nbytes, hipMemcpyHostToDevice, st));
    GHIP(hipMemsetAsync((char*)buf + nbytes, 0, nbytes, st));
    if (g->transport == FGPU_TRANSPORT_RCCL) {
        GNCCL(g->rccl.GroupStart());
        GNCCL(g->rccl.Send(buf, (size_t)nbytes, ncclUint8, rank, r.comm, st));
        GNCCL(g->rccl.Recv((char*)buf + nbytes, (size_t)nbytes, ncclUint8, rank, r.comm, st));
        GNCCL(g->rccl.GroupEnd());
    } else {
        // the copy transport's path with both ends on this rank: post, take, copy, settle
        hipEvent_t ready;
        if (int rc = new_event(g, rank, &ready)) return rc;
        GHIP(hipEventRecord(ready, st));
        Msg *msg = nullptr, *got = nullptr;
        if (int rc = post(g, rank, rank, buf, nbytes, ready, false, &msg)) return rc;
        if (int rc = take(g, rank, rank, nbytes, &got)) return rc;
        if (int rc = copy_in(g, rank, got, (char*)buf + nbytes)) return rc;
        if (int rc = settle(g, rank, msg)) return rc;
    }
    GHIP(hipMemcpyAsync(back.data(), (char*)buf + nbytes, nbytes, hipMemcpyDeviceToHost, st));
    GHIP(hipStreamSynchronize(st));
    *ok = memcmp(src.data(), back.data(), (size_t)nbytes) == 0 ? 1 : 0;
    return FGPU_OK;
}

// ---- device memory for the host's exchange buffers ---------------------------------------------------------------------------------------
int fgpu_device_alloc(fgpu_ctx* ctx, uint64_t nbytes, void** dptr) {
    if (!ctx || !dptr) return FGPU_ERR_ARG;
    *dptr = nullptr;
    FGPU_HIP(hipSetDevice(ctx->prm.device));
    hipError_t e = hipMalloc(dptr, nbytes ? nbytes : 16);
    if (e != hipSuccess) {
        *dptr = nullptr;
        ctx->err = std::string("fgpu_device_alloc: hipMalloc of ") + std::to_string(nbytes) + " bytes failed: " + hipGetErrorString(e);
        return FGPU_ERR_NOMEM;
    }
    return FGPU_OK;
}

int fgpu_device_free(fgpu_ctx* ctx, void* dptr) {
    if (!ctx) return FGPU_ERR_ARG;
    if (!dptr) return FGPU_OK;
    FGPU_HIP(hipSetDevice(ctx->prm.device));
    FGPU_HIP(hipStreamSynchronize(ctx->stream));
    FGPU_HIP(hipFree(dptr));
    return FGPU_OK;
}

int fgpu_device_copy(fgpu_ctx* ctx, void* dst_dev, const void* src_dev, uint64_t nbytes) {
    if (!ctx || (nbytes && (!dst_dev || !src_dev))) return FGPU_ERR_ARG;
    FGPU_HIP(hipSetDevice(ctx->prm.device));
    if (nbytes) FGPU_HIP(hipMemcpyAsync(dst_dev, src_dev, nbytes, hipMemcpyDeviceToDevice, ctx->stream));
    return FGPU_OK;
}

int fgpu_device_zero(fgpu_ctx* ctx, void* dst_dev, uint64_t nbytes) {
    if (!ctx || (nbytes && !dst_dev)) return FGPU_ERR_ARG;
    FGPU_HIP(hipSetDevice(ctx->prm.device));
    if (nbytes) FGPU_HIP(hipMemsetAsync(dst_dev, 0, nbytes, ctx->stream));
    return FGPU_OK;
}

int fgpu_scan_pairs_devptr(fgpu_ctx* ctx, int which, void** dptr, uint64_t* nbytes) {
    if (!ctx || !dptr || (which != 0 && which != 1)) return FGPU_ERR_ARG;
    if (which == 0) {
        if (!ctx->short_pf) { ctx->err = "fgpu_scan_pairs_devptr: no short pair filter on the device (fgpu_scan_short_pairs)"; return FGPU_ERR_STATE; }
        *dptr = ctx->short_pf;
        if (nbytes) *nbytes = ctx->short_pf_tai / 8;
    } else {
        if (ctx->lp.mode != FGPU_LONG_PAIRS_FILTER || !ctx->lp.bits) { ctx->err = "fgpu_scan_pairs_devptr: no long pair filter on the device (fgpu_scan_long_pairs, FGPU_LONG_PAIRS_FILTER)"; return FGPU_ERR_STATE; }
        *dptr = ctx->lp.bits;
        if (nbytes) *nbytes = ctx->lp.tai / 8;
    }
    return FGPU_OK;
}

}  // extern "C"
