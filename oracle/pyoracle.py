"""ctypes view of oracle/liboracle.so (the C restatement in mdbg_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg as the checker -- never by metamdbg_amd (the product path).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
REFDRV = os.path.join(_HERE, "_ref", "refdrv")


def build(with_ref: bool = True) -> None:
    subprocess.run(["make", "-C", _HERE, "liboracle.so"], check=True, stdout=subprocess.DEVNULL)
    if with_ref and os.path.isdir("/root/reference/src"):
        subprocess.run(["make", "-C", _HERE, "ref"], check=True, stdout=subprocess.DEVNULL)


class ReadRecord(C.Structure):
    _fields_ = [("n", C.c_uint32), ("minimizers", C.POINTER(C.c_uint32)), ("pos", C.POINTER(C.c_uint32)),
                ("dir", C.POINTER(C.c_uint8)), ("qual", C.POINTER(C.c_uint8)), ("mean_quality", C.c_float),
                ("read_length", C.c_uint32), ("hpc_length", C.c_uint32), ("low_complexity", C.c_int),
                ("low_quality", C.c_int)]


class ScanParams(C.Structure):
    _fields_ = [("K", C.c_uint), ("density", C.c_float), ("hpc", C.c_int), ("min_read_quality", C.c_float),
                ("repetitive", C.POINTER(C.c_uint32)), ("n_rep", C.c_size_t)]


class KminmerTable(C.Structure):
    _fields_ = [("n", C.c_uint64), ("k", C.c_uint), ("vecs", C.POINTER(C.c_uint32)),
                ("hash_lo", C.POINTER(C.c_uint64)), ("hash_hi", C.POINTER(C.c_uint64)),
                ("abundance", C.POINTER(C.c_uint32)), ("n_solid", C.c_uint64)]


class AbundanceMap(C.Structure):
    _fields_ = [("n", C.c_uint64), ("hi", C.POINTER(C.c_uint64)), ("lo", C.POINTER(C.c_uint64)),
                ("abundance", C.POINTER(C.c_uint32))]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build(with_ref=False)
        L = C.CDLL(_LIB_PATH)
        L.orc_kmer_hash.restype = C.c_uint64
        L.orc_kmer_hash.argtypes = [C.c_uint64]
        L.orc_density_threshold.restype = C.c_uint64
        L.orc_density_threshold.argtypes = [C.c_float]
        L.orc_hpc_encode.restype = C.c_size_t
        L.orc_sequence_complexity.restype = C.c_double
        L.orc_sequence_complexity.argtypes = [C.c_char_p, C.c_size_t]
        L.orc_mean_read_quality.restype = C.c_float
        L.orc_mean_read_quality.argtypes = [C.c_char_p, C.c_size_t]
        L.orc_read_selection.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(ScanParams), C.POINTER(ReadRecord)]
        L.orc_write_read_record.restype = C.c_size_t
        L.orc_compute_n50.restype = C.c_uint32
        L.orc_correction_scan.argtypes = L.orc_read_selection.argtypes
        L.orc_apply_density_threshold.argtypes = [C.c_void_p, C.c_size_t, C.c_float, C.c_void_p]
        L.orc_apply_density_threshold.restype = C.c_size_t
        L.orc_compute_n50.argtypes = [C.c_void_p, C.c_size_t]
        L.orc_compute_mean_length.restype = C.c_uint32
        L.orc_compute_mean_length.argtypes = [C.c_void_p, C.c_size_t]
        L.orc_compute_last_k.restype = C.c_int
        L.orc_compute_last_k.argtypes = [C.c_float, C.c_size_t, C.c_size_t, C.c_size_t]
        L.orc_purge_palindrome.restype = C.c_size_t
        L.orc_purge_palindrome.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t]
        L.orc_kminmer_normalize.restype = C.c_int
        L.orc_kminmer_normalize.argtypes = [C.c_void_p, C.c_uint, C.c_void_p]
        L.orc_kminmer_hash128.argtypes = [C.c_void_p, C.c_uint, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.orc_kminmer_count_first.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint, C.c_uint32, C.POINTER(KminmerTable)]
        L.orc_kminmer_count_refined.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint, C.POINTER(AbundanceMap), C.POINTER(KminmerTable)]
        L.orc_kminmer_index.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint, C.POINTER(AbundanceMap), C.POINTER(KminmerTable)]
        L.orc_small_contigs.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint, C.c_uint, C.POINTER(AbundanceMap), C.c_void_p]
        L.orc_small_contigs.restype = None
        L.orc_abundance_map_from_records.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(AbundanceMap)]
        L.orc_abundance_map_overlay.argtypes = [C.POINTER(AbundanceMap), C.c_void_p, C.c_uint32, C.c_uint, C.c_uint32]
        L.orc_murmur3_x64_128.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p]
        _lib = L
    return _lib


def kmer_hash(v: int) -> int:
    return lib().orc_kmer_hash(v & 0xFFFFFFFFFFFFFFFF)


def density_threshold(density: float) -> int:
    return lib().orc_density_threshold(density)


def murmur128(data: bytes, seed: int = 0) -> tuple[int, int]:
    out = (C.c_uint64 * 2)()
    lib().orc_murmur3_x64_128(data, len(data), seed, out)
    return out[0], out[1]


def minimizer_parse(seq: bytes, K: int, density: float, hpc: bool, trim: int = 1, repetitive=None):
    """EncoderRLE + MinimizerParser::parse with an explicit _trimBps: (values, positions, directions) as lists."""
    L = lib()
    L.orc_minimizer_parse_trim.restype = C.c_size_t
    rle = C.create_string_buffer(len(seq) + 2)
    rpos = (C.c_uint64 * (len(seq) + 2))()
    hl = L.orc_hpc_encode(seq, C.c_size_t(len(seq)), int(hpc), rle, rpos)
    rep = np.ascontiguousarray(repetitive if repetitive is not None else [], dtype=np.uint32)
    om = (C.c_uint32 * max(hl, 1))(); op = (C.c_uint32 * max(hl, 1))(); od = (C.c_uint8 * max(hl, 1))()
    n = L.orc_minimizer_parse_trim(rle, C.c_size_t(hl), K, C.c_float(density), rep.ctypes.data_as(C.POINTER(C.c_uint32)),
                                   C.c_size_t(len(rep)), C.c_size_t(trim), om, op, od)
    return list(om[:n]), list(op[:n]), list(od[:n])


def correction_scan(seq: bytes, qual: bytes | None, K: int = 13, density: float = 0.025, hpc: bool = False,
                    repetitive=None) -> dict:
    """One read -> what ReadCorrection::ReadSelectionFunctor hands to its record sink."""
    return read_selection(seq, qual, K, density, hpc, 0.0, repetitive, _fn="orc_correction_scan")


def apply_density_threshold(mins, density: float) -> np.ndarray:
    """Indices kept by Utils::applyDensityThreshold."""
    m = np.ascontiguousarray(mins, dtype=np.uint32)
    keep = np.zeros(len(m), dtype=np.uint8)
    lib().orc_apply_density_threshold(m.ctypes.data, len(m), C.c_float(density), keep.ctypes.data)
    return np.flatnonzero(keep)


def read_selection(seq: bytes, qual: bytes | None, K: int = 15, density: float = 0.005, hpc: bool = True,
                   min_read_quality: float = 0.0, repetitive=None, _fn: str = "orc_read_selection") -> dict:
    """One read -> the record the reference's readSelection would write for it."""
    rep = np.ascontiguousarray(repetitive if repetitive is not None else [], dtype=np.uint32)
    p = ScanParams(K, density, int(hpc), min_read_quality,
                   rep.ctypes.data_as(C.POINTER(C.c_uint32)), len(rep))
    rec = ReadRecord()
    getattr(lib(), _fn)(seq, qual, len(seq), C.byref(p), C.byref(rec))
    n = rec.n
    out = dict(
        minimizers=np.ctypeslib.as_array(rec.minimizers, (n,)).copy() if n else np.zeros(0, np.uint32),
        pos=np.ctypeslib.as_array(rec.pos, (n,)).copy() if n else np.zeros(0, np.uint32),
        dir=np.ctypeslib.as_array(rec.dir, (n,)).copy() if n else np.zeros(0, np.uint8),
        qual=np.ctypeslib.as_array(rec.qual, (n,)).copy() if n else np.zeros(0, np.uint8),
        mean_quality=rec.mean_quality, read_length=rec.read_length, hpc_length=rec.hpc_length,
        low_complexity=bool(rec.low_complexity), low_quality=bool(rec.low_quality))
    buf = (C.c_uint8 * (13 + 10 * n))()
    nb = lib().orc_write_read_record(C.byref(rec), buf)
    out["record"] = bytes(buf[:nb])
    lib().orc_read_record_free(C.byref(rec))
    return out


def purge_palindrome(mins, first_k: int, last_k: int) -> np.ndarray:
    m = np.ascontiguousarray(mins, dtype=np.uint32).copy()
    n = lib().orc_purge_palindrome(m.ctypes.data, len(m), first_k, last_k)
    return m[:n]


def kminmer_normalize_hash(vec) -> tuple[int, np.ndarray, int, int]:
    v = np.ascontiguousarray(vec, dtype=np.uint32)
    out = np.empty_like(v)
    rev = lib().orc_kminmer_normalize(v.ctypes.data, len(v), out.ctypes.data)
    hi, lo = C.c_uint64(), C.c_uint64()
    lib().orc_kminmer_hash128(out.ctypes.data, len(v), C.byref(hi), C.byref(lo))
    return rev, out, hi.value, lo.value


def _table_to_numpy(t: KminmerTable) -> dict:
    n, k = t.n, t.k
    d = dict(
        n=n, k=k, n_solid=t.n_solid,
        hash_lo=np.ctypeslib.as_array(t.hash_lo, (n,)).copy() if n else np.zeros(0, np.uint64),
        hash_hi=np.ctypeslib.as_array(t.hash_hi, (n,)).copy() if n else np.zeros(0, np.uint64),
        abundance=np.ctypeslib.as_array(t.abundance, (n,)).copy() if n else np.zeros(0, np.uint32),
        vecs=None)
    if t.vecs:
        d["vecs"] = np.ctypeslib.as_array(t.vecs, (n * k,)).copy().reshape(n, k) if n else np.zeros((0, k), np.uint32)
    lib().orc_kminmer_table_free(C.byref(t))
    return d


def kminmer_count_first(mins, offsets, k: int, min_abundance: int = 0) -> dict:
    m = np.ascontiguousarray(mins, dtype=np.uint32)
    o = np.ascontiguousarray(offsets, dtype=np.uint64)
    t = KminmerTable()
    lib().orc_kminmer_count_first(m.ctypes.data, o.ctypes.data, len(o) - 1, k, min_abundance, C.byref(t))
    return _table_to_numpy(t)


class PrevAbundance:
    """Previous-iteration abundance map (CreateMdbg::loadRefinedAbundances)."""

    def __init__(self, abundance_records: bytes):
        self.m = AbundanceMap()
        lib().orc_abundance_map_from_records(abundance_records, len(abundance_records) // 20, C.byref(self.m))

    def overlay_unitigs(self, unitigs: list[tuple[np.ndarray, int]], kprev: int) -> None:
        for mins, a in unitigs:
            u = np.ascontiguousarray(mins, dtype=np.uint32)
            lib().orc_abundance_map_overlay(C.byref(self.m), u.ctypes.data, len(u), kprev, a)
        lib().orc_abundance_map_finish(C.byref(self.m))

    def arrays(self):
        n = self.m.n
        if not n:
            return np.zeros(0, np.uint64), np.zeros(0, np.uint64), np.zeros(0, np.uint32)
        return (np.ctypeslib.as_array(self.m.hi, (n,)).copy(), np.ctypeslib.as_array(self.m.lo, (n,)).copy(),
                np.ctypeslib.as_array(self.m.abundance, (n,)).copy())

    def __del__(self):
        try:
            lib().orc_abundance_map_free(C.byref(self.m))
        except Exception:
            pass


def kminmer_count_refined(mins, offsets, k: int, prev: PrevAbundance) -> dict:
    m = np.ascontiguousarray(mins, dtype=np.uint32)
    o = np.ascontiguousarray(offsets, dtype=np.uint64)
    t = KminmerTable()
    lib().orc_kminmer_count_refined(m.ctypes.data, o.ctypes.data, len(o) - 1, k, C.byref(prev.m), C.byref(t))
    return _table_to_numpy(t)


def kminmer_index(mins, offsets, k: int, prev: PrevAbundance) -> dict:
    m = np.ascontiguousarray(mins, dtype=np.uint32)
    o = np.ascontiguousarray(offsets, dtype=np.uint64)
    t = KminmerTable()
    lib().orc_kminmer_index(m.ctypes.data, o.ctypes.data, len(o) - 1, k, C.byref(prev.m), C.byref(t))
    return _table_to_numpy(t)


def small_contigs(mins, offsets, k: int, kprev: int, prev: PrevAbundance) -> np.ndarray:
    """1 per sequence that IndexKminmerFunctor writes to smallContigs_k<k>.bin (graph/CreateMdbg.hpp:1330-1352)."""
    m = np.ascontiguousarray(mins, dtype=np.uint32)
    o = np.ascontiguousarray(offsets, dtype=np.uint64)
    flags = np.zeros(len(o) - 1, np.uint8)
    lib().orc_small_contigs(m.ctypes.data, o.ctypes.data, len(o) - 1, k, kprev, C.byref(prev.m), flags.ctypes.data)
    return flags


def edge_index(vecs) -> tuple[np.ndarray, np.ndarray, int]:
    """(hi, lo, checksum) of the distinct prefix/suffix identities of k-min-mer vectors (EdgeIndexer)."""
    v = np.ascontiguousarray(vecs, dtype=np.uint32)
    n, k = v.shape
    hi = np.zeros(2 * n, np.uint64); lo = np.zeros(2 * n, np.uint64)
    ck = C.c_uint64()
    f = lib().orc_edge_index
    f.restype = C.c_uint64
    f.argtypes = [C.c_void_p, C.c_uint64, C.c_uint, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
    m = f(v.ctypes.data, n, k, hi.ctypes.data, lo.ctypes.data, C.byref(ck))
    return hi[:m].copy(), lo[:m].copy(), ck.value


def unitig_edge_index(mins, offsets, k: int) -> tuple[np.ndarray, np.ndarray, int]:
    """(hi, lo, checksum) of the distinct unitig-edge identities, sorted by (hi, lo)."""
    m = np.ascontiguousarray(mins, dtype=np.uint32)
    o = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = len(o) - 1
    hi = np.zeros(max(4 * n, 1), dtype=np.uint64)
    lo = np.zeros(max(4 * n, 1), dtype=np.uint64)
    ck = C.c_uint64()
    f = lib().orc_unitig_edge_index
    f.restype = C.c_uint64
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
    d = f(m.ctypes.data, o.ctypes.data, n, k, hi.ctypes.data, lo.ctypes.data, C.byref(ck))
    return hi[:d].copy(), lo[:d].copy(), ck.value


def table_abundance_records(t: dict) -> np.ndarray:
    """20-byte records (lo, hi, abundance) of an oracle table, as a structured array."""
    from metamdbg_amd.formats import ABUNDANCE_DTYPE
    a = np.empty(t["n"], dtype=ABUNDANCE_DTYPE)
    a["lo"], a["hi"], a["abundance"] = t["hash_lo"], t["hash_hi"], t["abundance"]
    return a
