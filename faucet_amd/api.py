"""Host-side mirror of the reference interface for the hot path, on top of the C ABI.

Names follow the reference so that the parity tests read like its own:

    Bloom.create_bloom_filter_optimal / create_bloom_filter_2_hash     utils/Bloom.cpp:206-247
    load_two_filters(bloo1, bloo2, reads)                               utils/Bloom.cpp:267-350
    ReadScanner(...).scanReads(reads) / printScanSummary fields         src/ReadScanner.cpp:19-27,284-359
    JunctionMap.writeToFile                                             utils/JunctionMap.cpp:579-596

All compute runs in libfaucet_gpu.so on the MI355X; nothing here computes k-mers, hashes or junctions on the
host, and nothing falls back to a CPU path: without the library or without a gfx950 device the calls raise.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


class FaucetGpuError(RuntimeError):
    pass


def _check(rc, ctx=None):
    if rc != L.OK:
        msg = L.load().fgpu_last_error(ctx).decode(errors="replace") if ctx is not None else L.load().fgpu_last_error(None).decode()
        raise FaucetGpuError(f"libfaucet_gpu error {rc}: {msg}")


# ---- sizing (src/Faucet.cpp:197-219) --------------------------------------------------------------
def solve_p1(estimated_kmers: int, singletons: int, fp: float = 0.04) -> float:
    it = C.c_int32(0)
    p1 = L.load().fgpu_solve_p1(estimated_kmers, singletons, C.c_float(fp), C.byref(it))
    if p1 < 0:
        raise ValueError("p1 solver: root not bracketed (singletons must be > 0 and < estimated_kmers)")
    return p1


def size_optimal(estimated: int, fp: float):
    b, t, h = C.c_int32(), C.c_uint64(), C.c_int32()
    L.load().fgpu_size_optimal(estimated, C.c_float(fp), C.byref(b), C.byref(t), C.byref(h))
    return b.value, t.value, h.value


def size_two_hash(estimated: int, fp: float):
    b, t, h = C.c_int32(), C.c_uint64(), C.c_int32()
    L.load().fgpu_size_two_hash(estimated, C.c_float(fp), C.byref(b), C.byref(t), C.byref(h))
    return b.value, t.value, h.value


def load_filter_shape(estimated_kmers: int, singletons: int, fp: float = 0.04):
    """(tai, n_hash) of bloo1/bloo2 as getBloomFilterFromReads sizes them (src/Faucet.cpp:204-219)."""
    p1 = solve_p1(estimated_kmers, singletons, fp)
    _, tai, nh = size_optimal(estimated_kmers, np.float32(p1))
    return tai, nh


# ---- read batches -----------------------------------------------------------------------------------
class ReadBatch:
    """Sequence lines in file order.  Host numpy arrays or device pointers (e.g. torch tensors' data_ptr())."""

    def __init__(self, bases, offsets, n_reads=None, on_device=False, keepalive=None, starts=None, n_positions=None):
        self.on_device = bool(on_device)
        self.n_positions = n_positions          # bases + one separator per read, when the caller knows it (device batches)
        self._keep = (bases, offsets, keepalive)
        self.starts_ptr = int(starts) if starts else None      # device batches whose reads lie inside raw text
        if on_device:
            self.bases_ptr, self.offsets_ptr = int(bases), int(offsets)
            self.n_reads = int(n_reads)
        else:
            self.bases = np.ascontiguousarray(bases, dtype=np.uint8)
            self.offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
            self.bases_ptr = self.bases.ctypes.data
            self.offsets_ptr = self.offsets.ctypes.data
            self.n_reads = len(self.offsets) - 1
            self.n_positions = int(self.offsets[-1] - self.offsets[0]) + self.n_reads

    @classmethod
    def from_lines(cls, lines):
        offs = np.zeros(len(lines) + 1, dtype=np.uint64)
        if lines:
            offs[1:] = np.cumsum([len(x) for x in lines], dtype=np.uint64)
        data = b"".join(lines)
        bases = np.frombuffer(data, dtype=np.uint8).copy() if data else np.zeros(1, np.uint8)
        return cls(bases, offs)

    @classmethod
    def from_matrix(cls, mat: np.ndarray):
        n, ln = mat.shape
        return cls(np.ascontiguousarray(mat).reshape(-1), np.arange(n + 1, dtype=np.uint64) * np.uint64(ln))

    def c_struct(self):
        # device batches: the total the caller knows (n_positions = bases + one separator per read) spares the library a read-back of two
        # offsets, i.e. a wait for everything it has queued, per call
        total = self.n_positions - self.n_reads if (self.on_device and self.n_positions is not None) else 0
        return L.Reads(self.bases_ptr, self.offsets_ptr, self.n_reads, int(self.on_device), total if 0 < total < 1 << 32 else 0, self.starts_ptr)


STOP_DTYPE = np.dtype([("ext", "<u8"), ("read", "<u4"), ("info", "<u4")])
JUNC_DTYPE = np.dtype([("cov", np.uint8, 4), ("dist", np.uint8, 5), ("linked", np.uint8, 5)])
NEIGHBOR_DTYPE = np.dtype([("kmer", "<u8"), ("dist", "<i4"), ("len", "<i4"), ("node", "i1"), ("rindex", "i1"), ("abort", "i1"), ("reserved", "i1"),
                           ("reserved2", "<i4")])   # fgpu_neighbor


class _PinnedMemory:
    """The owner of one fgpu_host_alloc block: the ONLY place that frees it, and only when the last reference is gone -- the HostBuffer
    that made it, or any numpy view handed out over it (every view's base chain ends in a ctypes array that holds this object)."""

    def __init__(self, lib, nbytes):
        self.lib = lib
        self.ptr = lib.fgpu_host_alloc(nbytes)
        if not self.ptr:
            raise FaucetGpuError("fgpu_host_alloc failed")

    def __del__(self):
        ptr, self.ptr = getattr(self, "ptr", None), None
        if ptr:
            self.lib.fgpu_host_free(ptr)


class HostBuffer:
    """Page-locked host memory (fgpu_host_alloc) seen as numpy arrays: device-to-host copies into it run at link speed and asynchronously.
    Views keep the memory alive (a view that outlived its buffer pointed into hipHostFree'd memory): free() and __del__ only
    drop THIS object's reference, the block goes back when the last view is gone too."""

    def __init__(self, nbytes: int):
        self.lib = L.load()
        self.nbytes = int(nbytes)
        self._mem = _PinnedMemory(self.lib, max(self.nbytes, 1))
        self.ptr = self._mem.ptr
        self._raw = (C.c_uint8 * max(self.nbytes, 1)).from_address(self.ptr)
        self._raw._owner = self._mem          # numpy view -> .base = this ctypes array -> the block's owner

    def view(self, dtype=np.uint8, count=None) -> np.ndarray:
        if self._raw is None:
            raise FaucetGpuError("HostBuffer.view() after free()")
        return np.frombuffer(self._raw, dtype=dtype, count=-1 if count is None else count)

    def free(self):
        """give this object's reference up; the memory itself is released once no view of it is left"""
        self._raw = self._mem = self.ptr = None

    def __del__(self):
        self.free()


class Context:
    """One fgpu_ctx: one MI355X, one pair of load filters, one junction map."""

    def __init__(self, k, tai, n_hash, j=1, max_spacer_dist=100, device=0, profile=False, junction_capacity=0,
                 max_batch_bases=0, stream=None, walk_window_span=0, eager_flags=False, keep_resident=True, record_stops=False, mercy=False,
                 key_order_from_start=False):
        self.lib = L.load()
        flags = ((L.FLAG_PROFILE if profile else 0) | (L.FLAG_EAGER_FLAGS if eager_flags else 0)
                 | (0 if keep_resident else L.FLAG_NO_RESIDENT) | (L.FLAG_RECORD_STOPS if record_stops else 0)
                 | (L.FLAG_MERCY if mercy else 0) | (L.FLAG_KEY_ORDER_FROM_START if key_order_from_start else 0))
        p = L.Params(k, j, max_spacer_dist, n_hash, tai, device, flags, junction_capacity,
                     max_batch_bases, stream, walk_window_span)
        h = C.c_void_p()
        _check(self.lib.fgpu_create(C.byref(p), C.byref(h)))
        self.h = h
        self.k, self.tai, self.n_hash, self.j = k, tai, n_hash, j
        self.mercy = bool(mercy)
        self.stream = int(stream) if stream else None    # the caller's stream the library runs on (None: a stream of its own)
        self.device = int(device)

    def close(self):
        if getattr(self, "h", None):
            self.lib.fgpu_destroy(self.h)
            self.h = None
        for b in getattr(self, "_pinned", {}).values():
            b.free()
        self._pinned = {}

    def __del__(self):
        self.close()

    def _c(self, rc):
        _check(rc, self.h)

    # pass 1
    def load_begin(self, keep_carry=False, shard_times=False, shard_planes=False):
        self._c(self.lib.fgpu_load_begin(self.h, (L.LOAD_KEEP_CARRY if keep_carry else 0) | (L.LOAD_SHARD_TIMES if shard_times else 0) |
                                         (L.LOAD_SHARD_PLANES if shard_planes else 0)))

    def load_fixup(self, prefix_dev_ptr) -> dict:
        """multi-GPU: re-evaluate what this shard's own pass kept out of bloo2 against the OR of the lower ranks' bloo1"""
        st = L.LoadStats()
        self._c(self.lib.fgpu_load_fixup(self.h, prefix_dev_ptr, C.byref(st)))
        return st.as_dict()

    def load_fixup_state(self):
        """(can fgpu_load_fixup complete the load pass that has just ended?, bytes the context keeps load batches resident in)"""
        ready, budget = C.c_int(0), C.c_uint64(0)
        self._c(self.lib.fgpu_load_fixup_state(self.h, C.byref(ready), C.byref(budget)))
        return bool(ready.value), int(budget.value)

    def load_batch(self, batch: ReadBatch):
        s = batch.c_struct()
        self._c(self.lib.fgpu_load_batch(self.h, C.byref(s)))

    def presence_batch(self, batch: ReadBatch):
        s = batch.c_struct()
        self._c(self.lib.fgpu_presence_batch(self.h, C.byref(s)))

    def load_end(self) -> dict:
        st = L.LoadStats()
        self._c(self.lib.fgpu_load_end(self.h, C.byref(st)))
        return st.as_dict()

    def bloom_download(self, which=L.BLOO2) -> np.ndarray:
        out = np.empty(self.tai // 8, dtype=np.uint8)
        self._c(self.lib.fgpu_bloom_download(self.h, which, out.ctypes.data, out.nbytes))
        return out

    def bloom_download_begin(self, which, into: HostBuffer) -> np.ndarray:
        """Start the copy of a filter into page-locked memory; it runs beside whatever is submitted next (the scan).  The array
        returned is complete after bloom_download_wait()."""
        if into.nbytes < self.tai // 8:
            raise ValueError("host buffer smaller than the filter")
        self._c(self.lib.fgpu_bloom_download_begin(self.h, which, into.ptr, self.tai // 8))
        return into.view(np.uint8, self.tai // 8)

    def bloom_download_wait(self):
        self._c(self.lib.fgpu_bloom_download_wait(self.h))

    def _pinned_buffer(self, tag, nbytes) -> HostBuffer:
        if not hasattr(self, "_pinned"):
            self._pinned = {}
        b = self._pinned.get(tag)
        if b is None or b.nbytes < nbytes:
            # a tagged buffer is never regrown in place: the old block stays with whoever still holds a view of it (HostBuffer)
            b = self._pinned[tag] = HostBuffer(nbytes + nbytes // 4)
        return b

    def bloom_upload(self, which, data: np.ndarray):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        self._c(self.lib.fgpu_bloom_upload(self.h, which, data.ctypes.data, data.nbytes))

    def bloom_weight(self, which=L.BLOO2) -> float:
        w = C.c_float()
        self._c(self.lib.fgpu_bloom_weight(self.h, which, C.byref(w)))
        return w.value

    def bloom_devptr(self, which):
        p, n = C.c_void_p(), C.c_uint64()
        self._c(self.lib.fgpu_bloom_devptr(self.h, which, C.byref(p), C.byref(n)))
        return p.value, n.value

    def bitmap_or(self, dst_ptr, src_ptr, nbytes):
        self._c(self.lib.fgpu_bitmap_or(self.h, dst_ptr, src_ptr, nbytes))

    # pass 2
    def scan_begin(self):
        self._c(self.lib.fgpu_scan_begin(self.h))

    def scan_batch(self, batch: ReadBatch):
        s = batch.c_struct()
        self._c(self.lib.fgpu_scan_batch(self.h, C.byref(s)))

    def scan_prepare(self, batch: ReadBatch):
        s = batch.c_struct()
        self._c(self.lib.fgpu_scan_prepare(self.h, C.byref(s)))

    def scan_walk_prepared(self):
        self._c(self.lib.fgpu_scan_walk_prepared(self.h))

    def scan_end(self) -> dict:
        st = L.ScanStats()
        self._c(self.lib.fgpu_scan_end(self.h, C.byref(st)))
        return st.as_dict()

    def junctions(self, pinned=False):
        """(keys uint64[n], records[n]) in creation order.  pinned=True: into page-locked buffers owned by this context (the
        copy then runs at link speed); the arrays are valid until the next such call or close()."""
        n = C.c_uint64()
        self._c(self.lib.fgpu_scan_junction_count(self.h, C.byref(n)))
        cap = max(n.value, 1)
        if pinned:
            keys = self._pinned_buffer("jkeys", cap * 8).view(np.uint64, cap)
            recs = self._pinned_buffer("jrecs", cap * JUNC_DTYPE.itemsize).view(JUNC_DTYPE, cap)
        else:
            keys = np.zeros(cap, dtype=np.uint64)
            recs = np.zeros(cap, dtype=JUNC_DTYPE)
        got = C.c_uint64()
        self._c(self.lib.fgpu_scan_download_junctions(self.h, keys.ctypes.data, recs.ctypes.data, len(keys), C.byref(got)))
        return keys[: got.value], recs[: got.value]

    def table_entries(self) -> int:
        n = C.c_uint64()
        self._c(self.lib.fgpu_scan_table_entries(self.h, C.byref(n)))
        return n.value

    def export_table(self, dev_ptr, nbytes) -> int:
        n = C.c_uint64()
        self._c(self.lib.fgpu_scan_export_table(self.h, dev_ptr, nbytes, C.byref(n)))
        return n.value

    def import_hint(self, dev_ptr, n_entries):
        """a preview of the table this shard will be handed (an earlier state of it): lets scan_prepare evaluate lazily"""
        self._c(self.lib.fgpu_scan_import_hint(self.h, dev_ptr, n_entries))

    def import_table(self, dev_ptr, n_entries, carried: dict = None):
        st = None
        if carried is not None:
            st = L.ScanStats(**{k: int(v) for k, v in carried.items()})
        self._c(self.lib.fgpu_scan_import_table(self.h, dev_ptr, n_entries, C.byref(st) if st is not None else None))

    # probes / profiling
    def probe_hash(self, kmers):
        kmers = np.ascontiguousarray(kmers, dtype=np.uint64)
        n = len(kmers)
        c, a, b = (np.zeros(n, np.uint64) for _ in range(3))
        self._c(self.lib.fgpu_probe_hash(self.h, kmers.ctypes.data, n, c.ctypes.data, a.ctypes.data, b.ctypes.data))
        return c, a, b

    def probe_contains(self, which, canon):
        canon = np.ascontiguousarray(canon, dtype=np.uint64)
        out = np.zeros(len(canon), np.uint8)
        self._c(self.lib.fgpu_probe_contains(self.h, which, canon.ctypes.data, len(canon), out.ctypes.data))
        return out.astype(bool)

    def synchronize(self):
        self._c(self.lib.fgpu_synchronize(self.h))

    def text_reserve(self, max_chunk_bytes: int):
        """the largest chunk text_split will be handed: its buffers are sized for it once instead of growing with the first chunks"""
        self._c(self.lib.fgpu_text_reserve(self.h, int(max_chunk_bytes)))

    def text_split(self, text: bytes, fastq: bool, final_chunk: bool):
        """Record splitting of raw FASTA/FASTQ text on the device (the reference's getline loops).  Returns (ReadBatch that
        points into the device copy of the text and is valid until the next text_split, bytes consumed)."""
        buf = np.frombuffer(text, dtype=np.uint8) if len(text) else np.zeros(1, np.uint8)
        out, used = L.Reads(), C.c_uint64(0)
        self._c(self.lib.fgpu_text_split(self.h, buf.ctypes.data, len(text), 0, int(fastq), int(final_chunk), C.byref(out), C.byref(used)))
        rb = ReadBatch(out.bases or 0, out.offsets or 0, n_reads=int(out.n_reads), on_device=True, starts=out.starts)
        return rb, int(used.value)

    def scan_set_eager(self, on: bool):
        self._c(self.lib.fgpu_scan_set_eager(self.h, int(on)))

    def take_stops(self):
        """scanInputRead's lists of the next scanned batch: (batch number, structured array) or None when none is left"""
        n, seq = C.c_uint64(0), C.c_int64(0)
        rc = self.lib.fgpu_scan_take_stops(self.h, None, 0, C.byref(n), C.byref(seq))
        if rc not in (L.OK, L.ERR_CAPACITY):
            self._c(rc)
        if seq.value < 0:
            return None
        if rc == L.OK:                      # an empty batch fits any buffer and has been consumed
            return int(seq.value), np.zeros(0, dtype=STOP_DTYPE)
        out = np.zeros(max(int(n.value), 1), dtype=STOP_DTYPE)
        self._c(self.lib.fgpu_scan_take_stops(self.h, out.ctypes.data, len(out), C.byref(n), C.byref(seq)))
        return int(seq.value), out[:int(n.value)]

    def _probe_stage3(self, fn, kmers):
        kmers = np.ascontiguousarray(kmers, dtype=np.uint64)
        out = np.zeros(len(kmers), dtype=np.int8)
        self._c(fn(self.h, kmers.ctypes.data, len(kmers), out.ctypes.data))
        return out

    def probe_jcheck(self, kmers):                   # JChecker::jcheck(kmer_type), batched
        return self._probe_stage3(self.lib.fgpu_probe_jcheck, kmers)

    def probe_valid_extension(self, kmers):          # JunctionMap::getValidJExtension, batched
        return self._probe_stage3(self.lib.fgpu_probe_valid_extension, kmers)

    def probe_bloom_junction(self, kmers):           # JunctionMap::isBloomJunction, batched
        return self._probe_stage3(self.lib.fgpu_probe_bloom_junction, kmers)

    def diag_walk_probe(self):
        out = (C.c_uint64 * 4)()
        self._c(self.lib.fgpu_diag_walk_probe(self.h, out))
        return [int(v) for v in out]

    def scan_short_pairs(self, tai: int, n_hash: int, lists_to_host: bool = True):
        """keep the short pair filter (scan_forward's addPair rules) on the device from the next scan on; tai = 0 switches it off"""
        self._c(self.lib.fgpu_scan_short_pairs(self.h, int(tai), int(n_hash), 1 if lists_to_host else 0))

    def scan_short_pairs_download(self, tai: int):
        out = np.zeros(tai // 8, dtype=np.uint8)
        self._c(self.lib.fgpu_scan_short_pairs_download(self.h, out.ctypes.data, len(out)))
        return out

    def scan_long_pairs(self, tai: int, n_hash: int, mode: int = 2):
        """keep the long pair filter (scanReads' paired-end loop, src/ReadScanner.cpp:317-343) on the device from the next scan on;
        mode 0 = off, 1 = only the empty / not-empty pair counts (--no_cleaning), 2 = counts and filter"""
        self._c(self.lib.fgpu_scan_long_pairs(self.h, int(tai), int(n_hash), int(mode)))

    def scan_long_pairs_download(self, tai: int = 0):
        """(filter bytes or None, empty count, not-empty count) after scan_end"""
        out = np.zeros(tai // 8, dtype=np.uint8) if tai else None
        e, ne = C.c_uint64(0), C.c_uint64(0)
        self._c(self.lib.fgpu_scan_long_pairs_download(self.h, out.ctypes.data if tai else None, len(out) if tai else 0, C.byref(e), C.byref(ne)))
        return out, int(e.value), int(ne.value)

    def scan_pairs_devptr(self, which: int):
        """(device pointer, bytes) of a pair filter as it stands while a scan is open: 0 the short one, 1 the long one"""
        p, n = C.c_void_p(), C.c_uint64()
        self._c(self.lib.fgpu_scan_pairs_devptr(self.h, int(which), C.byref(p), C.byref(n)))
        return p.value, n.value

    def diag_long_pairs(self):
        out = (C.c_uint64 * 6)()
        self._c(self.lib.fgpu_diag_long_pairs(self.h, out))
        return dict(zip(("items", "paired_by_carry", "inserts", "rounds", "max_rounds", "batches"), (int(v) for v in out)))

    def diag_ovw(self):
        """the optimistic walk of large clusters during the last scan: pieces walked, rounds run, windows settled, windows left to the key-ordered walk"""
        out = (C.c_uint64 * 6)()
        self._c(self.lib.fgpu_diag_ovw(self.h, out))
        return dict(zip(("pieces", "rounds", "windows", "fallback_windows", "kept_piece_rounds", "table_overflow_windows"), (int(v) for v in out)))

    def scan_refresh_prepared(self):
        """the in-map planes of every prepared batch made again against the preview in the table (after a fresher import_hint)"""
        self._c(self.lib.fgpu_scan_refresh_prepared(self.h))

    def diag_prepared_refresh(self):
        out = (C.c_uint64 * 4)()
        self._c(self.lib.fgpu_diag_prepared_refresh(self.h, out))
        return dict(zip(("batches_in_full", "batches_merged", "new_keys", "mismatching_words"), (int(v) for v in out)))

    def diag_sparse_link(self):
        """windows of prepared batches (read shards) whose link pass visited candidate positions only / every position"""
        out = (C.c_uint64 * 2)()
        self._c(self.lib.fgpu_diag_sparse_link(self.h, out))
        return {"windows_sparse": int(out[0]), "windows_in_full": int(out[1])}

    def diag_ovw_tables(self):
        """event tables of the optimistic walk: the most entries a round of the last scan held, and the entries per table as they stand"""
        hw, cap = C.c_uint64(), C.c_uint64()
        self._c(self.lib.fgpu_diag_ovw_tables(self.h, C.byref(hw), C.byref(cap)))
        return {"high_water": int(hw.value), "capacity": int(cap.value)}

    def stage3_set_junctions(self, keys, recs):
        """the junction map Stage 3's walks look into (keys as JunctionMap keys them; records as junctions() returns them)"""
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        recs = np.ascontiguousarray(recs, dtype=JUNC_DTYPE)
        assert len(keys) == len(recs)
        self._c(self.lib.fgpu_stage3_set_junctions(self.h, keys.ctypes.data, recs.ctypes.data, len(keys)))

    def stage3_find_neighbors(self, start_kmers, indices, max_read_length: int, contigs: bool = False):
        """JunctionMap::findNeighbor for every (start k-mer, extension index) pair, whole walks on the device; returns (NEIGHBOR_DTYPE array,
        number of getValidJExtension evaluations) and, with contigs=True, the list of contig strings (BfSearchResult::contig) as third item"""
        start_kmers = np.ascontiguousarray(start_kmers, dtype=np.uint64)
        indices = np.ascontiguousarray(indices, dtype=np.int8)
        assert len(start_kmers) == len(indices)
        n = len(start_kmers)
        out = np.zeros(n, dtype=NEIGHBOR_DTYPE)
        probes = C.c_uint64(0)
        stride = int(self.lib.fgpu_stage3_contig_words(self.k, int(max_read_length))) if contigs else 0
        text = np.zeros(max(n * stride, 1), dtype=np.uint64)
        self._c(self.lib.fgpu_stage3_find_neighbors(self.h, start_kmers.ctypes.data, indices.ctypes.data, n, int(max_read_length), out.ctypes.data,
                                                    C.byref(probes), text.ctypes.data if contigs else None, stride))
        if not contigs:
            return out, int(probes.value)
        codes = (text.reshape(n, stride)[:, :, None] >> (2 * np.arange(32, dtype=np.uint64))[None, None, :]) & np.uint64(3)
        chars = np.frombuffer(b"ACTG", dtype=np.uint8)[codes.reshape(n, stride * 32).astype(np.int64)]
        return out, int(probes.value), [chars[i, :int(out["len"][i])].tobytes().decode() for i in range(n)]

    def kernel_times(self) -> dict:
        arr = (L.KernelTime * 64)()
        n = self.lib.fgpu_kernel_times(self.h, arr, 64)
        return {arr[i].name.decode(): (int(arr[i].launches), float(arr[i].total_ms)) for i in range(min(n, 64))}

    def profile_enable(self, on: bool):
        """HIP events around every kernel on / off (between passes)"""
        self._c(self.lib.fgpu_profile_enable(self.h, int(on)))

    def kernel_times_reset(self):
        self._c(self.lib.fgpu_kernel_times_reset(self.h))

    def diag_stream_copy(self, nbytes: int, iters: int = 5) -> float:
        """measured streaming-copy rate of this device, GB/s (read + write)"""
        out = C.c_double(0)
        self._c(self.lib.fgpu_diag_stream_copy(self.h, nbytes, iters, C.byref(out)))
        return out.value

    def diag_scan_replays(self) -> int:
        """how often the library has scanned a pass again by itself because the lazy junction-test preview could not be repaired"""
        n = C.c_uint64(0)
        self._c(self.lib.fgpu_diag_scan_replays(self.h, C.byref(n)))
        return int(n.value)

    def diag_late_flags(self) -> dict:
        """after scan_end: late junction tests that came out true at an unregistered k-mer, and how many of them voided the scan"""
        out = (C.c_uint64 * 3)()
        self._c(self.lib.fgpu_diag_late_flags(self.h, out))
        return {"noted": int(out[0]), "conflicts": int(out[1]), "swept": int(out[2])}

    def diag_load_split(self) -> dict:
        """where the last load pass settled its occurrences: routed to bloo2 by the marking kernel itself / left to the resolution"""
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._c(self.lib.fgpu_diag_load_split(self.h, C.byref(a), C.byref(b)))
        return {"in_mark": int(a.value), "pending": int(b.value)}

    def diag_binned_probes(self, table_bytes: int, n_probes: int, slice_bytes: int = 4 << 20, iters: int = 3) -> dict:
        """direct random bit probes vs probes binned by filter slice (NS1's query-side blocking), same addresses; rates in probes/s"""
        v = [C.c_double(0) for _ in range(4)]
        self._c(self.lib.fgpu_diag_binned_probes(self.h, table_bytes, n_probes, slice_bytes, iters, *[C.byref(x) for x in v]))
        return dict(zip(("direct_per_s", "binned_per_s", "bin_ms", "probe_ms"), [float(x.value) for x in v]))

    def diag_binned_chain(self, table_bytes: int, n_items: int, slice_bytes: int = 4 << 20, n_hash: int = 3, fill_byte: int = 0x29, iters: int = 3) -> dict:
        """Bloom::contains chains directly vs first level binned by filter slice + survivors handed back (NS1, whole chain); ms per pass"""
        v = [C.c_double(0) for _ in range(5)]
        eq = C.c_int(0)
        self._c(self.lib.fgpu_diag_binned_chain(self.h, table_bytes, n_items, slice_bytes, n_hash, fill_byte, iters, *[C.byref(x) for x in v], C.byref(eq)))
        d = dict(zip(("direct_ms", "bin_ms", "first_ms", "rest_ms", "survivors_share"), [float(x.value) for x in v]))
        d["equal"] = bool(eq.value)
        return d

    def diag_ko_trace(self) -> np.ndarray:
        """(n, 4) uint64: piece number, start, end, waited | lk positions << 48 of every piece the key-ordered walk walked (-DFGPU_KO_TRACE builds)"""
        out = np.zeros((1 << 21, 4), dtype=np.uint64)
        n = C.c_uint64(0)
        self._c(self.lib.fgpu_diag_ko_trace(self.h, out.ctypes.data, len(out), C.byref(n)))
        return out[: n.value].copy()

    def diag_ko_stamps(self) -> np.ndarray:
        """(pieces, 1024) uint64: per-step time stamps of one key-ordered piece in 16 (-DFGPU_KO_TRACE builds); word 0 = piece << 16 | stamps"""
        out = np.zeros((4096, 1024), dtype=np.uint64)
        n = C.c_uint64(0)
        self._c(self.lib.fgpu_diag_ko_stamps(self.h, out.ctypes.data, out.size, C.byref(n)))
        return out[: n.value].copy()

    def diag_device_attr(self) -> dict:
        v = [C.c_int32(0) for _ in range(4)]
        self._c(self.lib.fgpu_diag_device_attr(self.h, *[C.byref(x) for x in v]))
        return dict(zip(("memory_clock_khz", "memory_bus_bits", "l2_bytes", "compute_units"), [int(x.value) for x in v]))

    def diag_random_access(self, table_bytes: int, n_access: int, mode: int = 0, iters: int = 3) -> float:
        """measured independent random 32-bit accesses/s into a table of table_bytes (0 load, 1 atomicMin, 2 test+atomicOr)"""
        out = C.c_double(0)
        self._c(self.lib.fgpu_diag_random_access(self.h, table_bytes, n_access, mode, iters, C.byref(out)))
        return out.value


# ---- reference-shaped front end -------------------------------------------------------------------------------------
class Bloom:
    """Handle on one of the two load filters of a Context (utils/Bloom.h: tai, n_hash_func, blooma)."""

    def __init__(self, ctx: Context, which: int):
        self.ctx, self.which = ctx, which
        self.tai, self.n_hash_func = ctx.tai, ctx.n_hash

    def weight(self) -> float:                       # Bloom::weight
        return self.ctx.bloom_weight(self.which)

    def blooma(self) -> np.ndarray:                  # the raw bit array
        return self.ctx.bloom_download(self.which)

    def dump(self, path: str):                       # Bloom::dump (utils/Bloom.cpp:571-578)
        self.blooma().tofile(path)

    def load(self, path: str):                       # Bloom::load (utils/Bloom.cpp:580-587)
        self.ctx.bloom_upload(self.which, np.fromfile(path, dtype=np.uint8, count=self.tai // 8))

    def oldContains(self, canon_kmers):              # Bloom::oldContains (utils/Bloom.h:162-173), batched
        return self.ctx.probe_contains(self.which, canon_kmers)


def load_two_filters(bloo1: Bloom, bloo2: Bloom, batches) -> dict:
    """load_two_filters (utils/Bloom.cpp:267-350): `batches` is an iterable of ReadBatch in file order."""
    ctx = bloo1.ctx
    assert bloo2.ctx is ctx and bloo1.which == L.BLOO1 and bloo2.which == L.BLOO2
    ctx.load_begin()
    for b in batches:
        ctx.load_batch(b)
    return ctx.load_end()


class ReadScanner:
    """ReadScanner (src/ReadScanner.h:30-92) on the device junction map of a Context."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self.stats = None
        self.fell_back_to_eager = False

    def scanReads(self, batches):                    # src/ReadScanner.cpp:284-359
        replays = self.ctx.diag_scan_replays()
        st = self._scan(batches)
        # (the library scanned its journal again because the lazy junction-test preview could not be repaired: DESIGN.md section 4)
        self.fell_back_to_eager = self.ctx.diag_scan_replays() > replays
        return st

    def _scan(self, batches):
        self.ctx.scan_begin()
        for b in batches:
            self.ctx.scan_batch(b)
        self.stats = self.ctx.scan_end()
        return self.stats

    def junctions(self):
        return self.ctx.junctions()


_DEC = np.frombuffer(b"ACTG", dtype=np.uint8)


def print_kmer(kmer: int, k: int) -> str:            # print_kmer / code2seq (utils/Kmer.cpp:217)
    return "".join(chr(_DEC[(int(kmer) >> (2 * (k - 1 - i))) & 3]) for i in range(k))


def junction_lines(keys, recs, k):
    """The `.junctions` line of every record (Junction::toString, utils/Junction.cpp:74-89)."""
    out = []
    for key, r in zip(keys, recs):
        cv = [int(x) for x in r["cov"]]
        out.append("%s %s  %s  %s " % (print_kmer(int(key), k), " ".join(str(int(x)) for x in r["dist"]),
                                       " ".join(str(x) for x in cv + [sum(cv)]), " ".join(str(int(x)) for x in r["linked"])))
    return out


_ENC = {"A": 0, "C": 1, "T": 2, "G": 3}


def parse_junction_lines(lines, k):
    """Inverse of junction_lines: the `.junctions` text back to (keys, records), the way a restart with `-junctions_file` reads it
    (JunctionMap::buildFromFile utils/JunctionMap.cpp:619-639 + Junction(string) utils/Junction.cpp:102-118: k-mer word, 5 dists,
    4 coverages, one total that is skipped, 5 link flags, all whitespace-separated)."""
    lines = [ln for ln in lines if ln.strip()]
    keys = np.zeros(len(lines), dtype=np.uint64)
    recs = np.zeros(len(lines), dtype=JUNC_DTYPE)
    for i, ln in enumerate(lines):
        w = ln.split()
        if len(w) != 16 or len(w[0]) != k:
            raise ValueError(f"junction line {i}: expected a {k}-mer and 15 numbers, got {ln!r}")
        key = 0
        for ch in w[0]:
            key = (key << 2) | _ENC[ch]
        keys[i] = key
        v = [int(x) for x in w[1:]]
        recs[i]["dist"], recs[i]["cov"], recs[i]["linked"] = v[0:5], v[5:9], v[10:15]
    return keys, recs
