"""ctypes binding of oracle/liboracle.so (TEST INFRASTRUCTURE ONLY — see faucet_oracle.h).

Thin, explicit wrappers; numpy arrays in and out.  Builds the library with `make -C oracle` on
first use if it is missing (g++ only; no GPU, no reference needed).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")


def build(force: bool = False) -> str:
    if os.environ.get("FAUCET_ORACLE_LIB"):          # a sanitizer build of the same source (scripts/asan_oracle_fuzz.sh)
        return os.environ["FAUCET_ORACLE_LIB"]
    src_m = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("faucet_oracle.cpp", "faucet_oracle.h"))
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < src_m:
        subprocess.run(["make", "-C", _HERE, "liboracle.so"], check=True, capture_output=True)
    return _LIB_PATH


class LoadStats(C.Structure):
    _fields_ = [("reads_processed", C.c_uint64), ("unambiguous_reads", C.c_uint64), ("kmers", C.c_uint64),
                ("to_bloo2", C.c_uint64)]


class ScanStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("reads_processed", "unambiguous_reads", "reads_no_errors", "nb_jcheck_kmer",
                                          "nb_no_juncs", "nb_processed", "nb_skipped", "empty_count", "not_empty_count",
                                          "n_junctions")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class Reads(C.Structure):
    _fields_ = [("bases", C.c_void_p), ("offsets", C.c_void_p), ("n", C.c_uint64)]


JUNC_DTYPE = np.dtype([("cov", np.uint8, 4), ("dist", np.uint8, 5), ("linked", np.uint8, 5)])

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    u64, i32, vp, cp, f32, f64 = C.c_uint64, C.c_int, C.c_void_p, C.c_char_p, C.c_float, C.c_double
    sig = {
        "fo_nt2int": (i32, [C.c_char]), "fo_is_valid_nuc": (i32, [C.c_char]),
        "fo_encode": (u64, [cp, i32]), "fo_revcomp": (u64, [u64, i32]), "fo_canon": (u64, [u64, i32]),
        "fo_decode": (None, [u64, i32, cp]),
        "fo_seed": (u64, [i32]), "fo_old_hash": (u64, [u64, i32, u64]), "fo_bloom_tai": (u64, [u64]),
        "fo_solve_p1": (f64, [u64, u64, f32, C.POINTER(i32)]),
        "fo_size_optimal": (None, [u64, f32, C.POINTER(i32), C.POINTER(u64), C.POINTER(i32)]),
        "fo_size_two_hash": (None, [u64, f32, C.POINTER(i32), C.POINTER(u64), C.POINTER(i32)]),
        "fo_bloom_new": (vp, [u64, i32]), "fo_bloom_free": (None, [vp]), "fo_bloom_bits": (vp, [vp]),
        "fo_bloom_nbytes": (u64, [vp]), "fo_bloom_weight": (f32, [vp]), "fo_bloom_fakify": (None, [vp]),
        "fo_bloom_add_fake": (None, [vp, u64]), "fo_bloom_old_add": (None, [vp, u64]),
        "fo_bloom_old_contains": (i32, [vp, u64]), "fo_bloom_add_pair": (None, [vp, u64, u64, i32]),
        "fo_bloom_contains_pair": (i32, [vp, u64, u64, i32]), "fo_bloom_bit_tests": (u64, [vp]),
        "fo_bloom_bit_sets": (u64, [vp]), "fo_bloom_reset_counters": (None, [vp]),
        "fo_reads_from_file": (i32, [cp, i32, C.POINTER(Reads)]), "fo_reads_free": (None, [C.POINTER(Reads)]),
        "fo_stage3_jcheck": (i32, [vp, u64, i32, i32]),
        "fo_stage3_valid_extension": (i32, [vp, u64, i32, i32]),
        "fo_stage3_bloom_junction": (i32, [vp, u64, i32, i32]),
        "fo_load_two_filters": (None, [vp, vp, vp, vp, u64, i32, C.POINTER(LoadStats)]),
        "fo_load_two_filters_mercy": (None, [vp, vp, vp, vp, u64, i32, C.POINTER(LoadStats)]),
        "fo_load_single_filter": (None, [vp, vp, vp, u64, i32, C.POINTER(LoadStats)]),
        "fo_scanner_new": (vp, [i32, i32, i32, vp, vp, vp]), "fo_scanner_free": (None, [vp]),
        "fo_scan_reads": (None, [vp, vp, vp, u64, i32, i32]),
        "fo_scan_input_read": (u64, [vp, cp, u64, i32, vp, u64]),
        "fo_scan_get_stats": (None, [vp, C.POINTER(ScanStats)]),
        "fo_scan_get_junctions": (u64, [vp, i32, vp, vp, u64]),
        "fo_scan_write_junctions": (i32, [vp, cp]),
        "fo_scan_bit_tests_valid": (u64, [vp]),
        "fo_scan_import": (None, [vp, vp, vp, u64, C.POINTER(ScanStats)]),
        "fo_get_valid_reads": (u64, [vp, cp, u64, vp, u64]),
        "fo_test_for_junction": (i32, [vp, cp, u64, i32, C.POINTER(i32)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


# ------------------------------------------------------------------ helpers
def reads_from_lines(lines):
    """list of bytes -> (bases uint8[total], offsets uint64[n+1])"""
    offs = np.zeros(len(lines) + 1, dtype=np.uint64)
    if lines:
        offs[1:] = np.cumsum([len(x) for x in lines], dtype=np.uint64)
    bases = np.frombuffer(b"".join(lines), dtype=np.uint8).copy() if lines else np.zeros(0, np.uint8)
    return bases, offs


def reads_from_matrix(mat: np.ndarray):
    n, L = mat.shape
    return np.ascontiguousarray(mat).reshape(-1), (np.arange(n + 1, dtype=np.uint64) * np.uint64(L))


def reads_from_file(path: str, fastq: bool):
    r = Reads()
    if lib().fo_reads_from_file(path.encode(), int(fastq), C.byref(r)) != 0:
        raise FileNotFoundError(path)
    n = int(r.n)
    offs = np.ctypeslib.as_array(C.cast(r.offsets, C.POINTER(C.c_uint64)), shape=(n + 1,)).copy()
    total = int(offs[-1])
    bases = np.ctypeslib.as_array(C.cast(r.bases, C.POINTER(C.c_uint8)), shape=(max(total, 1),))[:total].copy()
    lib().fo_reads_free(C.byref(r))
    return bases, offs


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def solve_p1(E, S, fp=0.04):
    it = C.c_int(0)
    return lib().fo_solve_p1(E, S, C.c_float(fp), C.byref(it)), it.value


def size_optimal(E, fp):
    b, t, h = C.c_int(), C.c_uint64(), C.c_int()
    lib().fo_size_optimal(E, C.c_float(fp), C.byref(b), C.byref(t), C.byref(h))
    return b.value, t.value, h.value


def size_two_hash(E, fp):
    b, t, h = C.c_int(), C.c_uint64(), C.c_int()
    lib().fo_size_two_hash(E, C.c_float(fp), C.byref(b), C.byref(t), C.byref(h))
    return b.value, t.value, h.value


def sizing_from_cli(E, S, fp=0.04):
    """(tai, n_hash, p1, bits) of the two load filters exactly as src/Faucet.cpp:204-219 sizes them."""
    p1, _ = solve_p1(E, S, fp)
    bits, tai, nh = size_optimal(E, np.float32(p1))
    return tai, nh, p1, bits


class Bloom:
    def __init__(self, tai: int, n_hash: int):
        self.h = lib().fo_bloom_new(tai, n_hash)
        self.tai, self.n_hash = tai, n_hash

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:   # `lib` is already gone when the interpreter shuts down
            lib().fo_bloom_free(self.h)
            self.h = None

    def bits(self) -> np.ndarray:
        n = self.tai // 8
        return np.ctypeslib.as_array(C.cast(lib().fo_bloom_bits(self.h), C.POINTER(C.c_uint8)), shape=(n,))

    def set_bits(self, data: np.ndarray):
        self.bits()[:] = data

    def weight(self):
        return lib().fo_bloom_weight(self.h)

    def fakify(self, canon_kmers):
        lib().fo_bloom_fakify(self.h)
        for c in canon_kmers:
            lib().fo_bloom_add_fake(self.h, int(c))

    def contains(self, canon):
        return bool(lib().fo_bloom_old_contains(self.h, int(canon)))

    def add(self, canon):
        lib().fo_bloom_old_add(self.h, int(canon))

    def counters(self):
        return int(lib().fo_bloom_bit_tests(self.h)), int(lib().fo_bloom_bit_sets(self.h))

    def reset_counters(self):
        lib().fo_bloom_reset_counters(self.h)


def load_two_filters(bloo1: Bloom, bloo2: Bloom, bases, offs, k, mercy=False) -> LoadStats:
    st = LoadStats()
    fn = lib().fo_load_two_filters_mercy if mercy else lib().fo_load_two_filters
    fn(bloo1.h, bloo2.h, _p(bases), _p(offs), len(offs) - 1, k, C.byref(st))
    return st


def load_single_filter(bloo1: Bloom, bases, offs, k) -> LoadStats:
    st = LoadStats()
    lib().fo_load_single_filter(bloo1.h, _p(bases), _p(offs), len(offs) - 1, k, C.byref(st))
    return st


class Scanner:
    def __init__(self, k, j, max_spacer, bloom: Bloom, short_pf: Bloom = None, long_pf: Bloom = None):
        self._keep = (bloom, short_pf, long_pf)
        self.k = k
        self.h = lib().fo_scanner_new(k, j, max_spacer, bloom.h, short_pf.h if short_pf else None,
                                      long_pf.h if long_pf else None)

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:
            lib().fo_scanner_free(self.h)
            self.h = None

    def scan_reads(self, bases, offs, paired_ends=False, no_cleaning=True):
        lib().fo_scan_reads(self.h, _p(bases), _p(offs), len(offs) - 1, int(paired_ends), int(no_cleaning))

    def scan_input_read(self, line: bytes, no_cleaning=True):
        out = np.zeros(4096, dtype=np.uint64)
        n = lib().fo_scan_input_read(self.h, line, len(line), int(no_cleaning), _p(out), len(out))
        return out[:n].copy()

    def stats(self) -> dict:
        st = ScanStats()
        lib().fo_scan_get_stats(self.h, C.byref(st))
        return st.as_dict()

    def junctions(self, order="map"):
        n = int(lib().fo_scan_get_junctions(self.h, 0, None, None, 0))
        keys = np.zeros(n, dtype=np.uint64)
        recs = np.zeros(n, dtype=JUNC_DTYPE)
        lib().fo_scan_get_junctions(self.h, 0 if order == "map" else 1, _p(keys), _p(recs), n)
        return keys, recs

    def import_junctions(self, keys, recs, carried: dict = None):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        recs = np.ascontiguousarray(recs, dtype=JUNC_DTYPE)
        st = ScanStats(**{k: int(v) for k, v in carried.items()}) if carried else None
        lib().fo_scan_import(self.h, _p(keys), _p(recs), len(keys), C.byref(st) if st is not None else None)

    def bit_tests_valid(self) -> int:
        return int(lib().fo_scan_bit_tests_valid(self.h))

    def write_junctions(self, path: str):
        if lib().fo_scan_write_junctions(self.h, path.encode()) != 0:
            raise OSError(path)

    def valid_pieces(self, seg: bytes):
        out = np.zeros(2 * 1024, dtype=np.uint64)
        n = lib().fo_get_valid_reads(self.h, seg, len(seg), _p(out), 1024)
        return [(int(out[2 * i]), int(out[2 * i + 1])) for i in range(n)]

    def test_for_junction(self, piece: bytes, t: int):
        c = C.c_int(0)
        f = lib().fo_test_for_junction(self.h, piece, len(piece), t, C.byref(c))
        return bool(f), c.value


def decode(kmer: int, k: int) -> str:
    buf = C.create_string_buffer(k + 1)
    lib().fo_decode(int(kmer), k, buf)
    return buf.value.decode()


def junction_lines(keys, recs, k):
    """Render (keys, recs) in the .junctions text format (utils/Junction.cpp:74-89)."""
    out = []
    for key, r in zip(keys, recs):
        d = " ".join(str(int(x)) for x in r["dist"])
        cv = [int(x) for x in r["cov"]]
        c = " ".join(str(x) for x in cv + [sum(cv)])
        l = " ".join(str(int(x)) for x in r["linked"])
        out.append(f"{decode(int(key), k)} {d}  {c}  {l} ")
    return out
