"""Read-sharded two-pass pipeline over N processes (one per GPU): the exchange steps of DESIGN.md §5.

The reference is single-process; this is the multi-GPU host logic that SURVEY.md §8(e) asks for, kept apart from
`bench.py` so that the protocol itself (which collective, in which order, on which bitmap) is testable with the `gloo`
backend on CPU.  It talks to a *backend* object; the product backend is `GpuShard` below (libfaucet_gpu.so through
`api.Context`); the CPU tests plug in an oracle-backed stand-in with the same methods.

Pass 1 (exact), one of
    load of the shard alone -> exclusive prefix-OR of the shards' bloo1 -> fix-up against it -> bloo2 := OR over ranks
    presence bitmap of the shard  ->  exclusive prefix-OR over ranks = carried-in bloo1 of rank r
    ->  ordered load of the shard ->  bloo2 := OR over ranks
  Both exchanges work on SLICES of the bit arrays (RCCL has no bitwise-OR reduction): rank q collects slice q of every rank (grouped
  send/recv), reduces it with the local OR kernel -- a running OR for the exclusive prefix, whose r-th intermediate goes back to rank r --
  and the reduced slices are gathered.  Each rank receives 2 (N-1)/N of a bitmap per exchange instead of the N-1 bitmaps of an
  all-gather + local OR (0.9 instead of 3.5 GiB at N = 8 with 2^32-bit filters), and nothing waits on the host: the library runs on the
  stream the collectives are ordered with (GpuShard).
Pass 2:
    pure stage on every rank at once (no collective)  ->  the ordered junction walk is handed from rank to rank:
    recv table from r-1, import, walk own shard, export, send to r+1.  The last rank holds the final map.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from . import _lib as L

_STAT_NAMES = [n for n, _ in L.ScanStats._fields_]
HEADER_WORDS = 1 + len(_STAT_NAMES)          # [entries in the table, carried scan counters...]


# RCCL moves device tensors directly.  With the gloo backend (CPU tests; several ranks sharing one GPU when no multi-GPU box is
# at hand) device tensors are staged through host memory: same protocol, same library calls, another transport.
def _staged(t):
    return t.is_cuda and dist.get_backend() == "gloo"


class _Staging:
    """The host hop of the gloo transport, ordered the way ProcessGroupNCCL orders a collective: a transport stream of its own that
    waits (by event) for torch's CURRENT stream as of the call, and that the current stream waits for (by event) once data has landed.
    Page-locked bounce buffers, non-blocking copies; the host waits for exactly one thing -- the copy whose bytes gloo is about to send.
    Nothing here synchronises the device or the library's stream, so a library stream that is NOT the one the exchange is ordered with
    shows up as wrong bits in the multi-process tests, as it would under RCCL (`t.cpu()` used to hide that)."""

    def __init__(self):
        self._streams = {}

    def _stream(self, device):
        st = self._streams.get(device)
        if st is None:
            st = self._streams[device] = torch.cuda.Stream(device)
        return st

    def out(self, t):
        """device tensor -> page-locked host tensor, complete when this returns"""
        cur, st = torch.cuda.current_stream(t.device), self._stream(t.device)
        st.wait_stream(cur)
        h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        with torch.cuda.stream(st):
            h.copy_(t, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(st)
        ev.synchronize()
        return h

    @staticmethod
    def landing(t):
        return torch.empty(t.shape, dtype=t.dtype, pin_memory=True)

    def into(self, t, h):
        """page-locked host tensor -> device tensor; later work on the current stream is ordered behind the copy, the host does not wait"""
        cur, st = torch.cuda.current_stream(t.device), self._stream(t.device)
        st.wait_stream(cur)                  # whatever still reads or writes `t` on the current stream comes first
        with torch.cuda.stream(st):
            t.copy_(h, non_blocking=True)
        cur.wait_stream(st)


_staging = _Staging()


def _all_gather(out, t):
    if _staged(t):
        h = _staging.landing(out)
        dist.all_gather_into_tensor(h, _staging.out(t))
        _staging.into(out, h)
    else:
        dist.all_gather_into_tensor(out, t)


def _send(t, dst):
    dist.send(_staging.out(t) if _staged(t) else t, dst=dst)


def _isend(t, dst):
    """a send the caller does not wait for now: (request, the tensor that has to outlive it)"""
    src = _staging.out(t) if _staged(t) else t
    return dist.isend(src, dst=dst), src


def _recv(t, src):
    if _staged(t):
        h = _staging.landing(t)
        dist.recv(h, src=src)
        _staging.into(t, h)
    else:
        dist.recv(t, src=src)


class StageClock:
    """Per-rank stage times of one sharded step without stopping the device: an event on torch's current stream at every stage boundary
    (the stream the library and the collectives are ordered with, see GpuShard), read out after the step.  Off unless `enable`d (bench.py)."""

    def __init__(self):
        self.on, self.marks = False, []

    def enable(self, on=True):
        self.on, self.marks = bool(on), []

    def mark(self, name):
        if not self.on or not torch.cuda.is_available():
            return
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        self.marks.append((name, ev))

    def report(self):
        """[(stage, ms)]: time between the mark that opens a stage and the next one; call after a synchronisation"""
        out = [(b[0], float(a[1].elapsed_time(b[1]))) for a, b in zip(self.marks[:-1], self.marks[1:])]
        self.marks = []
        return out


CLOCK = StageClock()


def _agree(flag: bool, device=None) -> bool:
    """True iff it is True on every rank (the ranks must take the same protocol: their collectives have to match)"""
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device if dist.get_backend() != "gloo" else None)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item())


def load_sharded(backend, batches, rank: int, world: int):
    """Returns this shard's load stats; afterwards every rank holds the global bloo2.  Two exact protocols for pass 1:
    the fix-up one (no presence pass: every rank loads its shard alone, then re-evaluates what it kept out of bloo2 against the
    lower ranks' bits) where every rank can run it, else the presence one."""
    # Which one: the step waits for the SLOWEST rank, and under the presence protocol that is rank 0 -- it pays the presence pass AND the load
    # on an empty filter (every k-mer new: first-set-time atomics for all of them), while under the fix-up protocol every rank pays that
    # load once and the fix-up costs less than the presence pass.  Measured on one MI355X with one rank's shapes
    # (scripts/shard_load_times.py, scripts/shard_protocol_times.py), slowest rank, presence protocol vs fix-up protocol:
    #   config 4 on 8 ranks (25 M reads per rank, 2^33-bit filters): 162 + 358 = 520 ms  vs  358 + 83 = 441 ms  (a rank > 0: 162 + 256)
    #   weak shapes, 8 ranks (10 M reads per rank, 2^32 bits):        53 + 146 = 199 ms  vs  146 + 34 = 180 ms  (a rank > 0:  53 + 88)
    #   weak shapes, 2 ranks (2^30 bits):                              43 +  70 = 113 ms  vs   70 + 18 =  88 ms
    # Rounds 2-3 compared a rank > 0 only and took presence beyond 2 ranks.  So: the fix-up protocol wherever every rank can run it (the
    # shard's positions fit 32-bit times and its batches stay in HBM: config 4 from 8 ranks on; 4 ranks would need 5.05e9 times).
    # FAUCET_SHARD_PROTOCOL=fixup|presence overrides the choice.
    want = os.environ.get("FAUCET_SHARD_PROTOCOL", "auto")
    prefer = want != "presence"
    can = getattr(backend, "fixup_possible", None)
    if _agree(bool(prefer and can and can(batches)), getattr(backend, "device", None)):
        return load_sharded_fixup(backend, batches, rank, world)
    return load_sharded_presence(backend, batches, rank, world)


def _slices(nbytes: int, world: int):
    """[lo, hi) byte ranges of the `world` slices of a bitmap: equal, 16-byte aligned (the OR kernel's granule), the last one short"""
    step = -(-nbytes // world)
    step = (step + 15) & ~15
    return [(min(q * step, nbytes), min((q + 1) * step, nbytes)) for q in range(world)]


def _exchange(sends, recvs):
    """grouped point-to-point: sends = [(tensor, dst)], recvs = [(tensor, src)].  RCCL moves device tensors directly (one grouped
    launch: every rank talks to every other over its own xGMI link); gloo stages device tensors through host memory."""
    if not sends and not recvs:
        return
    ops, landing = [], []
    for t, dst in sends:
        ops.append(dist.P2POp(dist.isend, _staging.out(t) if _staged(t) else t, dst))
    for t, src in recvs:
        if _staged(t):
            h = _staging.landing(t)
            landing.append((t, h))
            ops.append(dist.P2POp(dist.irecv, h, src))
        else:
            ops.append(dist.P2POp(dist.irecv, t, src))
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    for t, h in landing:
        _staging.into(t, h)


def or_allreduce(backend, bitmap, rank: int, world: int):
    """bitmap := OR over ranks, in place on every rank: reduce-scatter by slices (grouped send/recv + local OR) + all-gather of the slices"""
    if world == 1:
        return
    nbytes = bitmap.numel()
    sl = _slices(nbytes, world)
    lo, hi = sl[rank]
    mine = hi - lo
    stage = backend.scratch(max(mine, 16) * (world - 1), tag="rs_stage")
    others = [q for q in range(world) if q != rank]
    backend.fence()      # (no-op when the library runs on the collectives' stream)
    # reduce-scatter: slice q of this rank's bitmap goes to rank q; slice `rank` of everybody else's comes here
    _exchange([(bitmap[sl[q][0]:sl[q][1]], q) for q in others if sl[q][1] > sl[q][0]],
              [(stage[i * mine:(i + 1) * mine], q) for i, q in enumerate(others)] if mine else [])
    backend.fence()      # what has just landed in `stage` (a copy torch queued) is read by the library's OR kernel next
    for i in range(len(others) if mine else 0):
        backend.or_tensor(bitmap[lo:hi], stage[i * mine:(i + 1) * mine])
    backend.fence()
    # all-gather: the reduced slice to everybody, theirs straight into place
    _exchange([(bitmap[lo:hi], q) for q in others] if mine else [],
              [(bitmap[sl[q][0]:sl[q][1]], q) for q in others if sl[q][1] > sl[q][0]])


def exclusive_prefix_or(backend, bitmap, out, rank: int, world: int):
    """out := OR of the bitmaps of all ranks < rank (zero on rank 0); `bitmap` is left as it was.  Slice q of every rank is collected on
    rank q, which ORs them up in rank order and sends the running value BEFORE rank r's contribution back to rank r."""
    nbytes = bitmap.numel()
    if world == 1:
        out.zero_()
        return
    sl = _slices(nbytes, world)
    lo, hi = sl[rank]
    mine = hi - lo
    others = [q for q in range(world) if q != rank]
    stage = backend.scratch(max(mine, 16) * world, tag="px_stage")      # slice `rank` of every rank, in rank order
    pref = backend.scratch(max(mine, 16) * world, tag="px_prefix")      # what goes back to rank r: OR over ranks < r of that slice
    backend.fence()
    if mine:
        stage[rank * mine:(rank + 1) * mine].copy_(bitmap[lo:hi])
    _exchange([(bitmap[sl[q][0]:sl[q][1]], q) for q in others if sl[q][1] > sl[q][0]],
              [(stage[q * mine:(q + 1) * mine], q) for q in others] if mine else [])
    if mine:
        pref[:mine].zero_()
        for r in range(1, world):                                        # running OR: pref[r] = pref[r-1] | stage[r-1]
            pref[r * mine:(r + 1) * mine].copy_(pref[(r - 1) * mine:r * mine])
            backend.fence()
            backend.or_tensor(pref[r * mine:(r + 1) * mine], stage[(r - 1) * mine:r * mine])
            backend.fence()
    # rank 0's prefix is empty: nothing is sent to it, it zero-fills
    sends = [(pref[q * mine:(q + 1) * mine], q) for q in others if q != 0] if mine else []
    recvs = [(out[sl[q][0]:sl[q][1]], q) for q in others if sl[q][1] > sl[q][0]] if rank != 0 else []
    _exchange(sends, recvs)
    if rank == 0:
        out.zero_()
    elif mine:
        out[lo:hi].copy_(pref[rank * mine:(rank + 1) * mine])
    backend.fence()


def load_sharded_fixup(backend, batches, rank: int, world: int):
    """own shard alone (first-set times kept)  ->  prefix = OR of the lower ranks' bloo1 (exclusive prefix-OR by slices)  ->  fix-up:
    an occurrence the local pass kept out of bloo2 goes there iff each of its bits is in the prefix or was set locally before it
    ->  bloo2 := OR over ranks.  Exact for the same reason as the presence protocol: what the sequential run has in
    bloo1 when it reaches shard r IS that prefix, and within the shard "set before t" is what the first-set times say."""
    CLOCK.mark("pass1_begin")
    stats = backend.load(batches, keep_carry=False, shard_times=True)
    CLOCK.mark("pass1_own_load")
    # can every rank complete its pass by the fix-up (all batches stayed resident)?  One "no" and ALL ranks run the presence protocol instead
    # (a pass more, the same filters) -- as host/shard_host.h does; backends without the query are taken at their word
    ready = getattr(backend, "fixup_ready", lambda: True)()
    if world > 1 and not _agree(bool(ready), getattr(backend, "device", None)):
        return load_sharded_presence(backend, batches, rank, world)
    b1 = backend.bloom_tensor(L.BLOO1)
    prefix = backend.scratch(b1.numel(), tag="prefix")
    exclusive_prefix_or(backend, b1, prefix, rank, world)
    CLOCK.mark("pass1_prefix_or_exchange")
    if rank > 0:
        stats = backend.load_fixup(prefix)
    CLOCK.mark("pass1_fixup")
    or_allreduce(backend, backend.bloom_tensor(L.BLOO2), rank, world)
    backend.fence()
    CLOCK.mark("pass1_or_allreduce")
    return stats


def load_sharded_presence(backend, batches, rank: int, world: int):
    """presence bitmap of the shard  ->  exclusive prefix-OR over ranks = carried-in bloo1 of rank r  ->  ordered load of the shard  ->
    bloo2 := OR over ranks"""
    CLOCK.mark("pass1_begin")
    backend.clear_filters()
    for b in batches:
        backend.presence(b)
    CLOCK.mark("pass1_presence")
    b1 = backend.bloom_tensor(L.BLOO1)
    prefix = backend.scratch(b1.numel(), tag="prefix")
    exclusive_prefix_or(backend, b1, prefix, rank, world)
    b1.copy_(prefix)
    backend.fence()      # the carried-in filter is in place before the library's load kernels read it
    CLOCK.mark("pass1_prefix_or_exchange")
    stats = backend.load(batches, keep_carry=True)
    CLOCK.mark("pass1_load")
    or_allreduce(backend, backend.bloom_tensor(L.BLOO2), rank, world)
    backend.fence()
    CLOCK.mark("pass1_or_allreduce")
    return stats


# share of the first shard's reads after which its table is shown to the other ranks.  Earlier means a thinner preview, later means more
# batches prepared without one: with the 2-rank shapes the pure stage of a later rank takes 91.8 ms behind a hint taken at 10 %, 75.0 ms
# at 25 % (117 without); 25 % arrives ~20 ms into the scan, 10 % ~8 ms (scripts/rank_stage_times.py, HINT_AFTER=...)
HINT_AFTER = 0.25
# Round 5: a rank that is handed the junction table passes it on at once, as a FRESHER PREVIEW, to the rank above, which makes the in-map planes of
# its prepared batches again against it while this rank walks; when its own table arrives it differs from that preview by what ONE shard created,
# and its walk looks only for those keys (faucet_gpu.h, fgpu_scan_refresh_prepared): 22 -> ~7 ms of every hop from the third rank on.
LATE_HINT = os.environ.get("FAUCET_LATE_HINT", "1") != "0"      # (has to be the same on every rank)
# Round 6: the rank above the first one has nobody to pass it a table, so the FIRST rank shows it a second, later state of its own table (after
# LATE_AFTER of its reads): that rank's planes and candidate planes are then made against a table a few million keys short of the one it is handed,
# and its hop is the short one every later rank has (config 4 on 8 shards: 94 -> 72 ms; the rest of the first shard's scan is the time it has).
LATE_AFTER = 0.7


def _n_reads(b):
    return b.n_reads if hasattr(b, "n_reads") else len(b[1]) - 1


def _bcast(t, src, rank):
    if _staged(t):
        h = _staging.out(t) if rank == src else _staging.landing(t)
        dist.broadcast(h, src=src)
        if rank != src:
            _staging.into(t, h)
    else:
        dist.broadcast(t, src=src)


def _bcast_async(t, src, rank):
    """the SENDER's side of a broadcast it does not wait for: (request, the tensor that has to outlive it).  The first rank shows its table in the
    middle of its scan; the receivers post the second of the two broadcasts only when their host has looked at the first (between two of their
    batches), and a broadcast the sender's stream waited for would hold its scan up for that long."""
    assert rank == src
    h = _staging.out(t) if _staged(t) else t
    return dist.broadcast(h, src=src, async_op=True), h


class _HintReceiver:
    """the two broadcasts of the hint (count, then entries) posted asynchronously and looked at between batches"""

    def __init__(self, backend, hdr, rank):
        self.backend, self.hdr, self.rank = backend, hdr, rank
        self.stage, self.buf, self.n = 0, None, 0
        self.work, self.host = self._post(hdr)

    def _post(self, t):
        if _staged(t):
            h = _staging.landing(t)
            return dist.broadcast(h, src=0, async_op=True), h
        return dist.broadcast(t, src=0, async_op=True), None

    def _landed(self, t):
        if self.host is not None:
            _staging.into(t, self.host)

    def _advance(self):
        if self.stage == 0:
            self._landed(self.hdr)
            self.n = int(self.hdr.cpu()[0])
            self.buf = self.backend.scratch(max(self.n, 1) * L.TABLE_ENTRY_BYTES, tag="table_hint")
            self.work, self.host = self._post(self.buf)
            self.stage = 1
        elif self.stage == 1:
            self._landed(self.buf)
            self.stage = 2
            return True
        return False

    def poll(self):
        while self.stage < 2 and self.work.is_completed():
            self.work.wait()
            if self._advance():
                self.backend.fence()
                self.backend.import_hint(self.buf, self.n)

    def finish(self):
        while self.stage < 2:
            self.work.wait()
            self._advance()


def _move_pair_filters(backend, rank: int, send: bool):
    """Both pair filters travel with the junction table (src/ReadScanner.cpp:208-225, 317-343): the short one only collects adds, the long one is
    check-then-insert in file order -- what shard r ends with is what shard r + 1 starts from.  Shards begin at even records (the caller's
    cut), so no first end waits across a cut."""
    for t in getattr(backend, "pair_tensors", lambda: [])():
        if send:
            _send(t, rank + 1)
        else:
            _recv(t, rank - 1)


def scan_sharded(backend, batches, rank: int, world: int):
    """Returns (stats, is_last): on the last rank the stats are the whole run's and backend.junctions() is the final map.
    A backend whose pair filters are on (GpuShard.pairs_setup before the call) has them handed from shard to shard with the table; the
    pair counts of the shards are the caller's to add up (backend.pair_counts()).

    The ranks behind the first one have no junction table while they run their pure stage, and with an empty table every junction test
    of every position is evaluated (105-116 ms per 10 M reads instead of ~50).  So the first rank shows them its table once it has
    walked HINT_AFTER of its reads -- an earlier state of the very table they will be handed, which is all the preview of the pure stage
    needs (faucet_gpu.h, fgpu_scan_import_hint) -- and they prepare against that.  The hint is replaced by the real table before a walk."""
    CLOCK.mark("pass2_begin")
    backend.scan_begin()
    hinting = world > 1
    hint_hdr = backend.header_tensor()[:1] if hinting else None
    late = LATE_HINT and hasattr(backend, "refresh_prepared")
    pending_sends = []
    if rank == 0:
        total, done, sent, sent_late = sum(_n_reads(b) for b in batches), 0, [False], [False]
        marks, marks_late = [], []
        for b in batches:
            done += _n_reads(b)
            marks.append(done >= HINT_AFTER * total)
            marks_late.append(done >= LATE_AFTER * total)
        hint_index = marks.index(True) if True in marks else len(batches) - 1
        late_index = marks_late.index(True) if True in marks_late else len(batches) - 1

        def show(i):
            if hinting and not sent[0] and i >= hint_index:
                sent[0] = True
                n, buf = backend.export_table(tag="table_hint_out")   # its own buffer: the broadcast may still be reading it at the final export
                backend.fence()
                hint_hdr[0] = n
                pending_sends.append(_bcast_async(hint_hdr, 0, rank))      # (round 6: not waited for here -- the scan goes on beside the broadcasts)
                pending_sends.append(_bcast_async(buf[:max(n, 1) * L.TABLE_ENTRY_BYTES], 0, rank))
            if hinting and late and not sent_late[0] and i >= late_index:
                # the fresher preview of the rank above (round 6): sent without waiting -- the scan goes on beside the copy
                sent_late[0] = True
                n, buf = backend.export_table(tag="table_late_out")
                backend.fence()
                lhdr = backend.header_tensor()
                lhdr[0] = n
                pending_sends.append(_isend(lhdr, 1))
                pending_sends.append(_isend(buf[:max(n, 1) * L.TABLE_ENTRY_BYTES], 1))

        # the first shard has nothing to wait for: it streams (pure stage of batch b+1 overlapped with the walk of batch b, lazy
        # junction tests), which puts the table on its way ~50 ms per 10 M reads earlier than prepare-all + walk
        stats = backend.scan_stream(batches, after_batch=show)   # ... and closes the pass
        show(len(batches))                                        # (a shard without batches still owes the others their broadcast)
        CLOCK.mark("pass2_first_shard_scan")
    else:
        # pure stage while the earlier shards walk.  It does not wait for the hint: the receive is posted, the batches prepared until it
        # has landed see an empty table (every test evaluated), the others the hint.  Measured with the per-rank shapes
        # (scripts/rank_stage_times.py, 10 M reads): 2 ranks 117 ms without a hint, 75 ms with it from the start; 8 ranks 107 and 93 ms.
        rx = _HintReceiver(backend, hint_hdr, rank) if hinting else None
        for b in batches:
            if rx:
                rx.poll()
            backend.scan_prepare(b)
        if rx:
            rx.finish()                                    # the collective is completed even when it came too late to be of use
        CLOCK.mark("pass2_pure_stage")
    hdr = backend.header_tensor()
    forwarded = None
    if rank >= 1 and late:
        # the table the rank below has just been handed: a preview one shard older than the table this rank will get (the rank above the
        # first one: the first rank's own table after LATE_AFTER of its reads)
        _recv(hdr, rank - 1)
        n_late = int(hdr.cpu().tolist()[0])
        late_buf = backend.scratch(max(n_late, 1) * L.TABLE_ENTRY_BYTES, tag="table_late")
        _recv(late_buf, rank - 1)
        backend.fence()
        backend.import_hint(late_buf, n_late)
        backend.refresh_prepared()
        CLOCK.mark("pass2_late_hint")
    if rank > 0:
        _recv(hdr, rank - 1)
        h = hdr.cpu().tolist()
        n_in = int(h[0])
        buf = backend.scratch(max(n_in, 1) * L.TABLE_ENTRY_BYTES, tag="table_in")
        _recv(buf, rank - 1)
        if late and rank < world - 1:                      # pass it on before walking on it (the sends complete beside the walk)
            fhdr = hdr.clone()
            forwarded = [_isend(fhdr, rank + 1), _isend(buf, rank + 1)]
        backend.fence()                                    # (scan_begin emptied the pair filters on the context's stream: not after they land)
        _move_pair_filters(backend, rank, send=False)
        backend.fence()
        CLOCK.mark("pass2_wait_for_table")                 # (the host waits here: the chain of walks of the lower ranks)
        carried = dict(zip(_STAT_NAMES, [int(x) for x in h[1:1 + len(_STAT_NAMES)]]))
        stats = backend.walk_shard(batches, buf, n_in, carried)   # import (replaces the hint) + ordered walk of this shard + scan_end
        CLOCK.mark("pass2_import_and_walk")
    for req, keep in (forwarded or []) + pending_sends:
        req.wait()
    if rank < world - 1:
        n_out, buf = backend.export_table()
        backend.fence()
        CLOCK.mark("pass2_export")
        hdr.zero_()
        hdr[0] = n_out
        hdr[1:1 + len(_STAT_NAMES)] = torch.tensor([stats[n] for n in _STAT_NAMES], dtype=torch.int64)
        _send(hdr, rank + 1)
        _send(buf, rank + 1)
        _move_pair_filters(backend, rank, send=True)
        CLOCK.mark("pass2_send")
    return stats, rank == world - 1


def _or_by_slices(backend, dst, src, world: int):
    """dst |= src, one slice of the N-rank layout after the other with the backend's OR kernel: the reduction step of the exchanges above"""
    for lo, hi in _slices(dst.numel(), world):
        if hi > lo:
            backend.or_tensor(dst[lo:hi], src[lo:hi])


def run_in_turn(make_backend, shards, protocol: str = "presence", after_load=None, after_scan=None):
    """The N-rank pipeline of this module executed by ONE process, one rank after the other -- for workloads whose N contexts do not fit one
    device together (BASELINE config 4: 8 shards x 32 GiB of first-set times) and for boxes without N GPUs.  Every rank runs the very
    backend calls it runs under load_sharded_* / scan_sharded, in a context of its own that is closed before the next one is made; what the
    ranks exchange travels as device tensors of this process, and the two reductions are the same slice-wise ORs (`_slices`, the backend's
    OR kernel).  Possible because every dependency of the protocol points from lower to higher ranks (the exclusive prefix-OR, the walk's
    hand-over) except the final OR of bloo2, which every scan needs: all loads come first, then all scans.

        make_backend()   -> a backend with GpuShard's methods (and close())
        shards           -> the ranks' batch lists, in file order
        protocol         -> "presence" | "fixup" (pass 1)
        after_load(r, stats, bloo1, bloo2): bloo1 / bloo2 = the SEQUENTIAL run's filters after shard r (tensors; valid during the call)
        after_scan(r, stats, backend):      backend.junctions() = the sequential run's map after shard r

    Returns (load stats per rank, scan stats of the last rank, the last rank's backend, still open)."""
    world = len(shards)
    close = lambda b: getattr(b, "close", lambda: None)()     # noqa: E731
    load_stats = []
    running = acc2 = None            # OR of the lower ranks' bloo1 bits (= the exclusive prefix of the next rank) / of the ranks' bloo2 so far
    if protocol == "presence":
        pres = []
        for r in range(world):
            b = make_backend()
            b.clear_filters()
            for batch in shards[r]:
                b.presence(batch)
            b.fence()
            pres.append(b.bloom_tensor(L.BLOO1).clone())
            close(b)
    for r in range(world):
        b = make_backend()
        b.clear_filters()
        b1 = b.bloom_tensor(L.BLOO1)
        if running is None:
            running, acc2 = torch.zeros_like(b1), torch.zeros_like(b1)
        if protocol == "presence":
            b1.copy_(running)            # carried-in bloo1 of rank r = exclusive prefix-OR of the presence bitmaps
            b.fence()
            stats = b.load(shards[r], keep_carry=True)
            b.fence()
            _or_by_slices(b, running, pres[r], world)
            pres[r] = None
            seq1 = b.bloom_tensor(L.BLOO1)                       # carried-in bits + this shard's = the sequential bloo1 after shard r
        elif protocol == "fixup":
            stats = b.load(shards[r], keep_carry=False, shard_times=True)
            b.fence()
            if r > 0:
                stats = b.load_fixup(running)
                b.fence()
            _or_by_slices(b, running, b.bloom_tensor(L.BLOO1), world)
            seq1 = running
        else:
            raise ValueError(protocol)
        _or_by_slices(b, acc2, b.bloom_tensor(L.BLOO2), world)
        b.fence()
        load_stats.append(stats)
        if after_load:
            after_load(r, stats, seq1, acc2)
        close(b)
    del running
    hint = [None, 0]
    table, n_table, stats = None, 0, None
    last = None
    older = None                     # the table the previous rank was handed (what it passes on as a fresher preview)
    pair_state = None                # the pair filters as the previous shard left them (pass 2; only with a backend whose filters are on)
    for r in range(world):
        b = make_backend()
        b.clear_filters()
        b.bloom_tensor(L.BLOO2).copy_(acc2)      # what the OR-allreduce leaves on every rank
        b.fence()
        b.scan_begin()
        if pair_state is not None:
            b.fence()                # (scan_begin empties the filters on the context's stream: not after the copies)
            for dst, src in zip(b.pair_tensors(), pair_state):
                dst.copy_(src)
            b.fence()
        if r == 0:
            total, done, marks = sum(_n_reads(x) for x in shards[0]), 0, []
            for x in shards[0]:
                done += _n_reads(x)
                marks.append(done >= HINT_AFTER * total)
            hint_index = marks.index(True) if True in marks else len(shards[0]) - 1
            done, marks_late = 0, []
            for x in shards[0]:
                done += _n_reads(x)
                marks_late.append(done >= LATE_AFTER * total)
            late_index = marks_late.index(True) if True in marks_late else len(shards[0]) - 1
            late0 = [None]

            def show(i, b=b):
                if hint[0] is None and i >= hint_index:
                    n, buf = b.export_table(tag="table_hint_out")
                    b.fence()
                    hint[0], hint[1] = buf[:max(n, 1) * L.TABLE_ENTRY_BYTES].clone(), n
                if late0[0] is None and i >= late_index and world > 1 and LATE_HINT and hasattr(b, "refresh_prepared"):
                    n, buf = b.export_table(tag="table_late_out")      # (round 6: the second rank's fresher preview, as scan_sharded sends it)
                    b.fence()
                    late0[0] = (buf[:max(n, 1) * L.TABLE_ENTRY_BYTES].clone(), n)

            stats = b.scan_stream(shards[0], after_batch=show)
            show(len(shards[0]))
            older = late0[0]
        else:
            b.import_hint(hint[0], hint[1])
            for batch in shards[r]:
                b.scan_prepare(batch)
            if older is not None and LATE_HINT and hasattr(b, "refresh_prepared"):      # the table the rank below was handed (rank 1: the first rank's late state), as scan_sharded passes it on
                b.import_hint(older[0], older[1])
                b.refresh_prepared()
            carried = {n: int(stats[n]) for n in _STAT_NAMES}
            stats = b.walk_shard(shards[r], table, n_table, carried)
            older = (table, n_table)
        if after_scan:
            after_scan(r, stats, b)
        if r < world - 1:
            n_table, buf = b.export_table()
            b.fence()
            table = buf[:max(n_table, 1) * L.TABLE_ENTRY_BYTES].clone()
            if getattr(b, "pair_tensors", None) and b.pair_tensors():
                pair_state = [t.clone() for t in b.pair_tensors()]
            close(b)
        else:
            last = b
    return load_stats, stats, last


class _DevView:
    """zero-copy torch view of device memory owned by libfaucet_gpu (through __cuda_array_interface__)"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class GpuShard:
    """The product backend: one api.Context on one MI355X.  The context should run on torch's current stream (api.Context(stream=
    torch.cuda.current_stream().cuda_stream), as bench.py creates it for N > 1): the library's kernels, the local OR steps and the
    collectives torch orders with that stream then follow each other without the host waiting in between.  A context on a stream of its
    own (stream_ordered=False) is fenced on the host around every exchange, as in round 1."""

    def __init__(self, ctx, device, stream_ordered=None):
        self.ctx, self.device = ctx, device
        self._scratch = {}
        self._forced = stream_ordered          # tests may force the fenced path; None = look at the streams themselves

    @property
    def stream_ordered(self) -> bool:
        """True iff the library's kernels and torch's copies / collectives are ordered by ONE stream: the context was created on a
        caller's stream and that stream is torch's current stream on this device right now.  Derived from the handles at every use
        (a hand-set attribute said nothing about the stream the collectives really run on)."""
        if self._forced is not None:
            return bool(self._forced)
        s = getattr(self.ctx, "stream", None)
        return s is not None and s == torch.cuda.current_stream(self.device).cuda_stream

    def fence(self):
        if self.stream_ordered:
            return
        self.ctx.synchronize()
        torch.cuda.synchronize(self.device)

    def close(self):
        self._scratch = {}
        self.ctx.close()

    def scratch(self, nbytes, tag="gather"):
        t = self._scratch.get(tag)
        if t is None or t.numel() < nbytes:
            t = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self._scratch[tag] = t
        return t[:nbytes]

    def header_tensor(self):
        return torch.zeros(HEADER_WORDS, dtype=torch.int64, device=self.device)

    def clear_filters(self):
        self.ctx.load_begin()
        self.ctx.load_end()

    def presence(self, batch):
        self.ctx.presence_batch(batch)

    def bloom_tensor(self, which):
        ptr, nbytes = self.ctx.bloom_devptr(which)
        return torch.as_tensor(_DevView(ptr, nbytes), device=self.device)

    def or_into(self, which, src):
        ptr, nbytes = self.ctx.bloom_devptr(which)
        self.ctx.bitmap_or(ptr, src.data_ptr(), nbytes)

    def _one_clock(self, batches):
        """does the fix-up protocol's own pass run on ONE 32-bit clock (True) or with fail planes (False)?  One rule for load() and
        fixup_possible(): the planes cover four hash functions, so FAUCET_SHARD_PLANES=1 (tests: planes at any size) only
        forces them where they exist."""
        pos = [getattr(b, "n_positions", None) for b in batches]
        fits = all(p is not None for p in pos) and sum(pos) + 64 * len(pos) < 0xFFF00000
        if fits and os.environ.get("FAUCET_SHARD_PLANES", "0") == "1" and getattr(self.ctx, "n_hash", 0) <= 4:
            return False
        return fits

    def load(self, batches, keep_carry, shard_times=False):
        # the fix-up protocol's own pass: on one clock where the shard's positions fit 32 bits (costs nothing), else with the fail planes
        short = self._one_clock(batches)
        self.ctx.load_begin(keep_carry=keep_carry, shard_times=shard_times and short, shard_planes=shard_times and not short)
        for b in batches:
            self.ctx.load_batch(b)
        return self.ctx.load_end()

    def fixup_possible(self, batches):
        """the fix-up protocol needs every batch of the shard kept in HBM with its fail planes (round 5: no 2^32-position limit any more)"""
        pos = [getattr(b, "n_positions", None) for b in batches]
        if any(p is None for p in pos) or getattr(self.ctx, "mercy", False):    # --mercy: fgpu_load_end leaves no fix-up state
            return False
        short = self._one_clock(batches)                 # one 32-bit clock for the shard; beyond: fail planes, which cover four hash functions
        if not short and getattr(self.ctx, "n_hash", 0) > 4:
            return False
        return sum(pos) + (64 << 20) < self.ctx.load_fixup_state()[1]     # 8 bits per position kept in HBM (codes, bad, sure, four fail planes) within the library's budget

    def fixup_ready(self):
        """after the own load of the fix-up protocol: did every batch stay resident (fgpu_load_fixup_state)?"""
        if os.environ.get("FAUCET_DEBUG_FIXUP_NOT_READY", "") == str(dist.get_rank() if dist.is_initialized() else 0):
            return False
        return self.ctx.load_fixup_state()[0]

    def or_tensor(self, dst, src):
        self.ctx.bitmap_or(dst.data_ptr(), src.data_ptr(), dst.numel())

    def load_fixup(self, prefix):
        return self.ctx.load_fixup(prefix.data_ptr())

    def pairs_setup(self, short=None, long=None, count_only=False):
        """the pair filters of the scans to come: short / long = (tai, n_hash) or None; count_only: the paired-end loop's two counts without
        the long filter (--no_cleaning).  The context needs record_stops."""
        self._pairs = [bool(short), bool(long)]
        self.ctx.scan_short_pairs(short[0], short[1], lists_to_host=False) if short else self.ctx.scan_short_pairs(0, 0, lists_to_host=False)
        if long:
            self.ctx.scan_long_pairs(long[0], long[1], 2)
        else:
            self.ctx.scan_long_pairs(0, 0, 1 if count_only else 0)
        self._long_tai = long[0] if long else 0

    def pair_tensors(self):
        """the filters that are on, as device tensors (valid from scan_begin on): what travels from shard to shard"""
        out = []
        for which, on in enumerate(getattr(self, "_pairs", [False, False])):
            if on:
                ptr, nbytes = self.ctx.scan_pairs_devptr(which)
                out.append(torch.as_tensor(_DevView(ptr, nbytes), device=self.device))
        return out

    def pair_counts(self):
        """(empty, not empty) pair counts of THIS shard's scan, after scan_end"""
        _, e, ne = self.ctx.scan_long_pairs_download(0)
        return e, ne

    def scan_begin(self):
        self.ctx.scan_begin()

    def scan_prepare(self, batch):
        self.ctx.scan_prepare(batch)

    def scan_stream(self, batches, after_batch=None):
        """scan_batch over all batches (after_batch(i) is called behind each), scan_end.  (A preview of the walk that does not hold is the
        library's business: it scans its journal again by itself; what has left this rank before that is at most a preview of the table,
        and the repeated scan's table is again a later state of it.)"""
        for i, b in enumerate(batches):
            self.ctx.scan_batch(b)
            if after_batch:
                after_batch(i)
        return self.ctx.scan_end()

    def import_hint(self, buf, n):
        self.ctx.import_hint(buf.data_ptr(), n)

    def refresh_prepared(self):
        self.ctx.scan_refresh_prepared()

    def walk_shard(self, batches, buf, n, carried):
        """the handed-over table takes the place of the preview, then the ordered walk of the prepared batches"""
        self.import_table(buf, n, carried)
        self.ctx.scan_walk_prepared()
        return self.ctx.scan_end()

    def scan_walk_prepared(self):
        self.ctx.scan_walk_prepared()

    def scan_end(self):
        return self.ctx.scan_end()

    def import_table(self, buf, n, carried):
        self.ctx.import_table(buf.data_ptr(), n, carried=carried)

    def export_table(self, tag="table_out"):
        n = self.ctx.table_entries()
        buf = self.scratch(max(n, 1) * L.TABLE_ENTRY_BYTES, tag=tag)
        got = self.ctx.export_table(buf.data_ptr(), buf.numel())
        assert got == n
        return n, buf

    def junctions(self):
        return self.ctx.junctions()
