"""ctypes loader for libfaucet_gpu.so (the C ABI of include/faucet_gpu.h).

Fails loudly: if the shared library is missing or does not export the ABI there is no fallback of any kind.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libfaucet_gpu.so")

OK, ERR_ARG, ERR_HIP, ERR_STATE, ERR_CAPACITY, ERR_NOMEM = range(6)
BLOO1, BLOO2 = 0, 1
FLAG_PROFILE = 1
FLAG_EAGER_FLAGS = 2
FLAG_NO_RESIDENT = 4
FLAG_RECORD_STOPS = 8
FLAG_MERCY = 16
FLAG_KEY_ORDER_FROM_START = 32
STOP_POS_MASK, STOP_FORWARD, STOP_FIRST, STOP_FAKE = 0x0FFFFFFF, 1 << 28, 1 << 29, 1 << 30
TABLE_ENTRY_BYTES = 32


class Params(C.Structure):
    _fields_ = [("k", C.c_int32), ("j", C.c_int32), ("max_spacer_dist", C.c_int32), ("n_hash", C.c_int32),
                ("tai", C.c_uint64), ("device", C.c_int32), ("flags", C.c_int32), ("junction_capacity", C.c_uint64),
                ("max_batch_bases", C.c_uint64), ("stream", C.c_void_p), ("walk_window_span", C.c_uint64)]


LOAD_KEEP_CARRY, LOAD_SHARD_TIMES, LOAD_SHARD_PLANES = 1, 2, 4


class Reads(C.Structure):
    _fields_ = [("bases", C.c_void_p), ("offsets", C.c_void_p), ("n_reads", C.c_uint64), ("on_device", C.c_int32),
                ("total_bases", C.c_uint32), ("starts", C.c_void_p)]


class LoadStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("reads_processed", "unambiguous_reads", "kmers", "to_bloo2")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class ScanStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("reads_processed", "unambiguous_reads", "reads_no_errors", "nb_jcheck_kmer", "nb_no_juncs",
                                          "nb_processed", "nb_skipped", "n_junctions", "kmers", "walk_windows", "walk_followers",
                                          "walk_max_cluster", "flag_positions", "piece_positions", "valid_reused", "flags_filled", "walk_parallel")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_uint64), ("total_ms", C.c_double)]


# every symbol include/faucet_gpu.h declares: (restype, argtypes)
_u64, _i32, _vp, _f32, _f64 = C.c_uint64, C.c_int32, C.c_void_p, C.c_float, C.c_double
_P = C.POINTER
SIGNATURES = {
    "fgpu_abi_version": (C.c_int, []),
    "fgpu_device_count": (C.c_int, []),
    "fgpu_create": (C.c_int, [_P(Params), _P(_vp)]),
    "fgpu_destroy": (None, [_vp]),
    "fgpu_last_error": (C.c_char_p, [_vp]),
    "fgpu_synchronize": (C.c_int, [_vp]),
    "fgpu_solve_p1": (_f64, [_u64, _u64, _f32, _P(_i32)]),
    "fgpu_bloom_tai": (_u64, [_u64]),
    "fgpu_size_optimal": (None, [_u64, _f32, _P(_i32), _P(_u64), _P(_i32)]),
    "fgpu_size_two_hash": (None, [_u64, _f32, _P(_i32), _P(_u64), _P(_i32)]),
    "fgpu_load_begin": (C.c_int, [_vp, C.c_int]),
    "fgpu_load_batch": (C.c_int, [_vp, _P(Reads)]),
    "fgpu_load_end": (C.c_int, [_vp, _P(LoadStats)]),
    "fgpu_presence_batch": (C.c_int, [_vp, _P(Reads)]),
    "fgpu_load_fixup": (C.c_int, [_vp, _vp, _P(LoadStats)]),
    "fgpu_load_fixup_state": (C.c_int, [_vp, _P(C.c_int), _P(C.c_uint64)]),
    "fgpu_scan_dump_order": (C.c_int, [_vp, _P(C.c_uint64), _P(C.c_uint64), C.c_uint64, C.c_uint64, _P(C.c_uint32)]),
    "fgpu_bloom_download": (C.c_int, [_vp, C.c_int, _vp, _u64]),
    "fgpu_bloom_download_begin": (C.c_int, [_vp, C.c_int, _vp, _u64]),
    "fgpu_bloom_download_wait": (C.c_int, [_vp]),
    "fgpu_bloom_upload": (C.c_int, [_vp, C.c_int, _vp, _u64]),
    "fgpu_bloom_weight": (C.c_int, [_vp, C.c_int, _P(_f32)]),
    "fgpu_bloom_devptr": (C.c_int, [_vp, C.c_int, _P(_vp), _P(_u64)]),
    "fgpu_bitmap_or": (C.c_int, [_vp, _vp, _vp, _u64]),
    "fgpu_host_alloc": (_vp, [_u64]),
    "fgpu_host_free": (None, [_vp]),
    "fgpu_text_reserve": (C.c_int, [_vp, _u64]),
    "fgpu_text_split": (C.c_int, [_vp, _vp, _u64, C.c_int, C.c_int, C.c_int, _P(Reads), _P(_u64)]),
    "fgpu_scan_begin": (C.c_int, [_vp]),
    "fgpu_scan_batch": (C.c_int, [_vp, _P(Reads)]),
    "fgpu_scan_prepare": (C.c_int, [_vp, _P(Reads)]),
    "fgpu_scan_walk_prepared": (C.c_int, [_vp]),
    "fgpu_scan_end": (C.c_int, [_vp, _P(ScanStats)]),
    "fgpu_scan_set_eager": (C.c_int, [_vp, C.c_int]),
    "fgpu_scan_take_stops": (C.c_int, [_vp, _vp, _u64, _P(_u64), _P(C.c_int64)]),
    "fgpu_scan_junction_count": (C.c_int, [_vp, _P(_u64)]),
    "fgpu_scan_download_junctions": (C.c_int, [_vp, _vp, _vp, _u64, _P(_u64)]),
    "fgpu_scan_table_entries": (C.c_int, [_vp, _P(_u64)]),
    "fgpu_scan_export_table": (C.c_int, [_vp, _vp, _u64, _P(_u64)]),
    "fgpu_scan_import_table": (C.c_int, [_vp, _vp, _u64, _P(ScanStats)]),
    "fgpu_scan_import_hint": (C.c_int, [_vp, _vp, _u64]),
    "fgpu_scan_short_pairs": (C.c_int, [_vp, _u64, _i32, _i32]),
    "fgpu_scan_short_pairs_download": (C.c_int, [_vp, _vp, _u64]),
    "fgpu_scan_long_pairs": (C.c_int, [_vp, _u64, _i32, _i32]),
    "fgpu_scan_long_pairs_download": (C.c_int, [_vp, _vp, _u64, _P(_u64), _P(_u64)]),
    "fgpu_diag_long_pairs": (C.c_int, [_vp, _P(_u64)]),
    "fgpu_diag_ovw": (C.c_int, [_vp, _P(_u64)]),
    "fgpu_diag_ovw_tables": (C.c_int, [_vp, _P(_u64), _P(_u64)]),
    "fgpu_scan_refresh_prepared": (C.c_int, [_vp]),
    "fgpu_diag_prepared_refresh": (C.c_int, [_vp, _P(_u64)]),
    "fgpu_diag_sparse_link": (C.c_int, [_vp, _P(_u64)]),
    "fgpu_probe_hash": (C.c_int, [_vp, _vp, _u64, _vp, _vp, _vp]),
    "fgpu_probe_contains": (C.c_int, [_vp, C.c_int, _vp, _u64, _vp]),
    "fgpu_probe_jcheck": (C.c_int, [_vp, _vp, _u64, _vp]),
    "fgpu_probe_valid_extension": (C.c_int, [_vp, _vp, _u64, _vp]),
    "fgpu_probe_bloom_junction": (C.c_int, [_vp, _vp, _u64, _vp]),
    "fgpu_stage3_set_junctions": (C.c_int, [_vp, _vp, _vp, _u64]),
    "fgpu_stage3_find_neighbors": (C.c_int, [_vp, _vp, _vp, _u64, _i32, _vp, _P(_u64), _vp, _u64]),
    "fgpu_stage3_contig_words": (_u64, [_i32, _i32]),
    "fgpu_kernel_times": (C.c_int, [_vp, _P(KernelTime), C.c_int]),
    "fgpu_kernel_times_reset": (C.c_int, [_vp]),
    "fgpu_profile_enable": (C.c_int, [_vp, C.c_int]),
    "fgpu_diag_stream_copy": (C.c_int, [_vp, _u64, C.c_int, _P(_f64)]),
    "fgpu_diag_random_access": (C.c_int, [_vp, _u64, _u64, C.c_int, C.c_int, _P(_f64)]),
    "fgpu_diag_ko_trace": (C.c_int, [_vp, _vp, _u64, _P(_u64)]),
    "fgpu_diag_ko_stamps": (C.c_int, [_vp, _vp, _u64, _P(_u64)]),
    "fgpu_diag_binned_chain": (C.c_int, [_vp, _u64, _u64, _u64, C.c_int, C.c_int, C.c_int, _P(_f64), _P(_f64), _P(_f64), _P(_f64), _P(_f64), _P(C.c_int)]),
    "fgpu_diag_device_attr": (C.c_int, [_vp, _P(_i32), _P(_i32), _P(_i32), _P(_i32)]),
    "fgpu_diag_walk_probe": (C.c_int, [_vp, _P(_u64)]),
    "fgpu_diag_load_split": (C.c_int, [_vp, _P(_u64), _P(_u64)]),
    "fgpu_diag_scan_replays": (C.c_int, [_vp, _P(_u64)]),
    "fgpu_diag_host_waits": (C.c_int, [_vp, _P(_u64), _P(_f64)]),
    "fgpu_diag_load_tables": (C.c_int, [_vp, _u64, _P(_f64), _P(_f64), _P(_f64)]),
    "fgpu_diag_pair_placements": (C.c_int, [_vp, C.c_int, _u64, _P(_f64)]),
    "fgpu_diag_late_flags": (C.c_int, [_vp, _P(_u64)]),
    "fgpu_diag_binned_probes": (C.c_int, [_vp, _u64, _u64, _u64, C.c_int, _P(_f64), _P(_f64), _P(_f64), _P(_f64)]),
    "fgpu_scan_pairs_devptr": (C.c_int, [_vp, C.c_int, _P(_vp), _P(_u64)]),
    "fgpu_device_alloc": (C.c_int, [_vp, _u64, _P(_vp)]),
    "fgpu_device_free": (C.c_int, [_vp, _vp]),
    "fgpu_device_copy": (C.c_int, [_vp, _vp, _vp, _u64]),
    "fgpu_device_zero": (C.c_int, [_vp, _vp, _u64]),
    "fgpu_group_create": (C.c_int, [C.c_int, C.c_int, _P(_vp)]),
    "fgpu_group_destroy": (None, [_vp]),
    "fgpu_group_attach": (C.c_int, [_vp, C.c_int, _vp]),
    "fgpu_group_abort": (None, [_vp]),
    "fgpu_group_last_error": (C.c_char_p, [_vp, C.c_int]),
    "fgpu_group_barrier": (C.c_int, [_vp, C.c_int]),
    "fgpu_group_or_allreduce": (C.c_int, [_vp, C.c_int, _vp, _u64]),
    "fgpu_group_exclusive_prefix_or": (C.c_int, [_vp, C.c_int, _vp, _vp, _u64]),
    "fgpu_group_send": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _u64]),
    "fgpu_group_send_async": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _u64]),
    "fgpu_group_flush": (C.c_int, [_vp, C.c_int]),
    "fgpu_group_recv": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _u64]),
    "fgpu_group_probe": (C.c_int, [_vp, C.c_int, C.c_int, _P(C.c_int), _P(_u64)]),
    "fgpu_group_selftest": (C.c_int, [_vp, C.c_int, _u64, _P(C.c_int)]),
}
TRANSPORT_COPY, TRANSPORT_RCCL = 0, 1

_lib = None


def load():
    """Load the shared library (once).  Raises if it is missing or incomplete — never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -m faucet_amd.build` (hipcc, gfx950). "
                           "There is no CPU fallback.")
    # One HIP runtime per process: PyTorch ships its own libamdhip64.so.7.  If torch is going to be used in this
    # process (bench.py, torch.distributed) it must be loaded FIRST so that our NEEDED libamdhip64.so.7 binds to
    # the copy already mapped; two runtimes in one process leave the second one without a device.
    if os.environ.get("FAUCET_NO_TORCH", "0") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    # FAUCET_GPU_LIB: another build of the SAME library (host code under AddressSanitizer: scripts/asan_gpu_fuzz.sh); never a fallback
    lib = C.CDLL(os.environ.get("FAUCET_GPU_LIB") or LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.fgpu_abi_version() != 3:
        raise RuntimeError("libfaucet_gpu.so ABI version mismatch")
    _lib = lib
    return lib
