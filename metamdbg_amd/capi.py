"""ctypes binding of libmdbg_hip.so (include/mdbg_hip.h) -- thin, no compute in Python.

There is no fallback: a missing library raises at import of :func:`lib`, and a missing or
non-gfx950 GPU raises :class:`MdbgError` from :class:`Context`.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MDBG_LIB") or os.path.join(_HERE, "libmdbg_hip.so")   # MDBG_LIB: experimental variant

MDBG_READ_LOW_COMPLEXITY = 1
MDBG_READ_LOW_QUALITY = 2


class MdbgError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libmdbg_hip error {code}: {msg}")
        self.code = code


class ScanParams(C.Structure):
    _fields_ = [("minimizer_size", C.c_uint32), ("density", C.c_float), ("hpc", C.c_int32),
                ("min_read_quality", C.c_float), ("repetitive", C.POINTER(C.c_uint32)),
                ("n_repetitive", C.c_uint32), ("apply_read_filters", C.c_int32), ("quality_window", C.c_int32), ("no_end_trim", C.c_int32),
                ("ignore_qualities", C.c_int32)]


# name -> (restype, argtypes); every symbol include/mdbg_hip.h declares
_P = C.c_void_p
_u64p = C.POINTER(C.c_uint64)
_u32p = C.POINTER(C.c_uint32)
SIGNATURES = {
    "mdbg_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "mdbg_destroy": (None, [_P]),
    "mdbg_last_error": (C.c_char_p, [_P]),
    "mdbg_synchronize": (C.c_int, [_P]),
    "mdbg_stream": (_P, [_P]),
    "mdbg_device_info": (C.c_int, [_P, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), _u64p]),
    "mdbg_set_option": (C.c_int, [_P, C.c_char_p, C.c_int64]),
    "mdbg_timing_enable": (C.c_int, [_P, C.c_int]),
    "mdbg_timing_reset": (C.c_int, [_P]),
    "mdbg_timing_get": (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_double), _u64p]),
    "mdbg_reads_from_ascii": (C.c_int, [_P, _P, _P, _P, C.c_uint32, C.POINTER(_P)]),
    "mdbg_reads_from_packed": (C.c_int, [_P, _P, _P, _P, C.c_uint32, C.POINTER(_P)]),
    "mdbg_reads_from_packed_async": (C.c_int, [_P, _P, _P, _P, C.c_uint32, C.POINTER(_P)]),
    "mdbg_reads_attach_qualities_async": (C.c_int, [_P, _P, C.c_char_p, _P]),
    "mdbg_reads_mark_ascii": (C.c_int, [_P, _P, _P, C.c_uint32, C.c_char_p, _P]),
    "mdbg_reads_wait": (C.c_int, [_P, _P]),
    "mdbg_reads_attach_qualities": (C.c_int, [_P, _P, C.c_char_p, _P]),
    "mdbg_reads_synthetic": (C.c_int, [_P, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, _P, _P, C.c_uint32,
                                       C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, C.POINTER(_P)]),
    "mdbg_reads_info": (C.c_int, [_P, _u32p, _u64p, _u64p]),
    "mdbg_reads_get": (C.c_int, [_P, _P, C.c_uint32, _P, _P, _u32p]),
    "mdbg_reads_export_ascii": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, _P, _P, _u64p]),
    "mdbg_reads_export_qualities": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, _P, _u64p]),
    "mdbg_memcpy_device": (C.c_int, [_P, _P, _P, C.c_uint64]),
    "mdbg_host_alloc": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "mdbg_host_free": (None, [_P, _P]),
    "mdbg_reads_free": (None, [_P]),
    "mdbg_scan": (C.c_int, [_P, _P, C.POINTER(ScanParams), C.POINTER(_P)]),
    "mdbg_minimizers_info": (C.c_int, [_P, _u32p, _u64p]),
    "mdbg_minimizers_to_host": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mdbg_minimizers_from_host": (C.c_int, [_P, _P, _P, C.c_uint32, C.POINTER(_P)]),
    "mdbg_minimizers_device_ptrs": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P)]),
    "mdbg_minimizers_free": (None, [_P]),
    "mdbg_minimizers_concat": (C.c_int, [_P, C.POINTER(_P), C.c_uint32, C.POINTER(_P)]),
    "mdbg_apply_density_threshold": (C.c_int, [_P, _P, C.c_float, C.POINTER(_P)]),
    "mdbg_purge_palindromes": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, C.POINTER(_P)]),
    "mdbg_repetitive_minimizers": (C.c_int, [_P, _P, _P, _u32p]),
    "mdbg_census_create": (C.c_int, [_P, C.POINTER(_P)]),
    "mdbg_census_add": (C.c_int, [_P, _P, _P]),
    "mdbg_census_top": (C.c_int, [_P, _P, _P, _u32p]),
    "mdbg_census_free": (None, [_P]),
    "mdbg_kminmer_count_first": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, C.POINTER(_P)]),
    "mdbg_prev_from_records": (C.c_int, [_P, _P, C.c_uint64, C.POINTER(_P)]),
    "mdbg_prev_overlay_unitigs": (C.c_int, [_P, _P, _P, _P, C.c_uint32]),
    "mdbg_kminmer_count_refined": (C.c_int, [_P, _P, _P, C.c_uint32, _P, C.POINTER(_P)]),
    "mdbg_kminmer_index": (C.c_int, [_P, _P, _P, C.c_uint32, _P, C.POINTER(_P)]),
    "mdbg_table_info": (C.c_int, [_P, _u32p, _u64p, _u64p, C.POINTER(C.c_int)]),
    "mdbg_table_to_host": (C.c_int, [_P, _P, _P, _P]),
    "mdbg_table_to_host_range": (C.c_int, [_P, _P, C.c_uint64, C.c_uint64, _P, _P]),
    "mdbg_table_checksum": (C.c_int, [_P, _P, _u64p]),
    "mdbg_table_stats": (C.c_int, [_P, _u64p]),
    "mdbg_first_pass_info": (C.c_int, [_P, _u64p]),
    "mdbg_stream_spin": (C.c_int, [_P, C.c_uint32]),
    "mdbg_minimizers_slice": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]),
    "mdbg_shard_exchange_local": (C.c_int, [_P, C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_void_p), _u64p, C.POINTER(C.c_void_p)]),
    "mdbg_device_clock_khz": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "mdbg_table_lookup": (C.c_int, [_P, _P, _P, _P, C.c_uint64, _P]),
    "mdbg_edge_index": (C.c_int, [_P, _P, C.POINTER(_P), _u64p]),
    "mdbg_unitig_edge_index": (C.c_int, [_P, _P, C.c_uint32, C.POINTER(_P), _u64p]),
    "mdbg_small_contigs": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, _P, _P]),
    "mdbg_table_keys_to_host": (C.c_int, [_P, _P, _P]),
    "mdbg_table_free": (None, [_P]),
    "mdbg_row_words": (C.c_uint32, [C.c_uint32]),
    "mdbg_shard_begin": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, C.POINTER(_P), C.POINTER(_P), _u64p]),
    "mdbg_shard_reduce": (C.c_int, [_P, _P, _P, C.c_uint64, C.POINTER(_P)]),
    "mdbg_shard_finish": (C.c_int, [_P, _P, _P, C.c_uint32, C.POINTER(_P)]),
    "mdbg_shard_free": (None, [_P]),
    "mdbg_shard_from_table": (C.c_int, [_P, _P, C.c_uint32, C.POINTER(_P), C.POINTER(_P), _u64p]),
    "mdbg_shard_keep": (C.c_int, [_P, _P, _P, C.POINTER(_P)]),
    "mdbg_comm_unique_id": (C.c_int, [_P]),
    "mdbg_comm_create": (C.c_int, [_P, _P, C.c_int, C.c_int, C.POINTER(_P)]),
    "mdbg_comm_create_mode": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "mdbg_comm_mode": (C.c_int, [_P]),
    "mdbg_comm_note": (C.c_char_p, [_P]),
    "mdbg_comm_times": (C.c_int, [_P, C.POINTER(C.c_double)]),
    "mdbg_comm_adopt": (C.c_int, [_P, _P, C.c_int, C.c_int, C.POINTER(_P)]),
    "mdbg_comm_destroy": (None, [_P]),
    "mdbg_comm_stats": (C.c_int, [_P, _u64p, C.POINTER(C.c_double)]),
    "mdbg_shard_abort": (C.c_int, [_P, _P, C.c_int]),
    "mdbg_kminmer_count_first_sharded": (C.c_int, [_P, _P, _P, C.c_uint32, C.c_uint32, C.POINTER(_P)]),
    "mdbg_shard_exchange": (C.c_int, [_P, _P, _P, _P, _u64p, C.POINTER(_P)]),
    "mdbg_bytes_create": (C.c_int, [_P, C.c_uint64, C.POINTER(_P)]),
    "mdbg_bytes_upload_async": (C.c_int, [_P, _P, C.c_uint64, _P, C.c_uint64, _u64p]),
    "mdbg_bytes_upload_done": (C.c_int, [_P, _P, C.c_uint64, C.c_int]),
    "mdbg_bytes_free": (None, [_P]),
    "mdbg_minimizers_from_record_bytes": (C.c_int, [_P, _P, _P, C.c_uint32, _P, C.POINTER(_P)]),
    "mdbg_prev_from_record_bytes": (C.c_int, [_P, _P, C.c_uint64, C.POINTER(_P)]),
}

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _ptr(a: np.ndarray | None):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Context:
    """One HIP device + stream (mdbg_create)."""

    def __init__(self, device: int = 0):
        self.h = C.c_void_p()
        rc = lib().mdbg_create(device, C.byref(self.h))
        if rc:
            raise MdbgError(rc, (lib().mdbg_last_error(None) or b"").decode())

    def check(self, rc: int) -> None:
        if rc:
            raise MdbgError(rc, (lib().mdbg_last_error(self.h) or b"").decode())

    def close(self) -> None:
        if self.h:
            lib().mdbg_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self) -> None:
        self.check(lib().mdbg_synchronize(self.h))

    def device_info(self) -> dict:
        arch = C.create_string_buffer(64)
        ncu, hbm = C.c_int(), C.c_uint64()
        self.check(lib().mdbg_device_info(self.h, arch, 64, C.byref(ncu), C.byref(hbm)))
        clk = C.c_int()
        lib().mdbg_device_clock_khz(self.h, C.byref(clk))
        return dict(arch=arch.value.decode(), n_cu=ncu.value, hbm_bytes=hbm.value, clock_khz=clk.value)

    def set_option(self, name: str, value: int) -> None:
        self.check(lib().mdbg_set_option(self.h, name.encode(), value))

    def stream_spin(self, microseconds: int) -> None:
        """One idle wave for that long on the context's stream (mdbg_stream_spin); returns at once."""
        self.check(lib().mdbg_stream_spin(self.h, microseconds))

    def first_pass_info(self) -> dict:
        """How the last kminmer_count_first of this context ran (mdbg_first_pass_info)."""
        a = (C.c_uint64 * 8)()
        self.check(lib().mdbg_first_pass_info(self.h, a))
        names = ("path", "groups", "bucket_bits", "levels", "attempts", "lds_slots", "buckets", "instances")
        return {k: int(v) for k, v in zip(names, a)}

    # -- timing ---------------------------------------------------------------------------
    def timing(self, on: bool) -> None:
        self.check(lib().mdbg_timing_enable(self.h, int(on)))

    def timing_reset(self) -> None:
        self.check(lib().mdbg_timing_reset(self.h))

    def timing_get(self, kernel: str) -> tuple[float, int]:
        ms, n = C.c_double(), C.c_uint64()
        self.check(lib().mdbg_timing_get(self.h, kernel.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    # -- reads ------------------------------------------------------------------------------
    def reads_from_ascii(self, seqs: list[bytes], quals: list[bytes] | None = None) -> "Reads":
        offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
        np.cumsum([len(s) for s in seqs], out=offs[1:])
        bases = b"".join(seqs)
        q = b"".join(quals) if quals is not None else None
        h = C.c_void_p()
        self.check(lib().mdbg_reads_from_ascii(self.h, bases, q, _ptr(offs), len(seqs), C.byref(h)))
        return Reads(self, h)

    def reads_from_packed(self, words: np.ndarray, word_off: np.ndarray, lens: np.ndarray, quals: list[bytes] | None = None) -> "Reads":
        words = np.ascontiguousarray(words, dtype=np.uint64)
        word_off = np.ascontiguousarray(word_off, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        h = C.c_void_p()
        self.check(lib().mdbg_reads_from_packed(self.h, _ptr(words), _ptr(word_off), _ptr(lens), len(lens), C.byref(h)))
        if quals is not None:
            offs = np.zeros(len(quals) + 1, dtype=np.uint64)
            np.cumsum([len(q) for q in quals], out=offs[1:])
            self.check(lib().mdbg_reads_attach_qualities(self.h, h, b"".join(quals), _ptr(offs)))
        return Reads(self, h)

    def reads_from_packed_async(self, words: np.ndarray, word_off: np.ndarray, lens: np.ndarray, quals: bytes | None = None,
                                qual_off: np.ndarray | None = None) -> "Reads":
        """The upload is queued on the context's upload stream and the call returns; the arrays must stay alive and untouched until
        reads.wait() (they are kept on the returned object).  Consumers (scan) order themselves after it on the device."""
        h = C.c_void_p()
        self.check(lib().mdbg_reads_from_packed_async(self.h, _ptr(words), _ptr(word_off), _ptr(lens), len(lens), C.byref(h)))
        r = Reads(self, h)
        r._keep = (words, word_off, lens, quals, qual_off)
        if quals is not None:
            self.check(lib().mdbg_reads_attach_qualities_async(self.h, h, quals, _ptr(qual_off)))
        return r

    def reads_synthetic(self, spec, first_read: int = 0, n_reads: int | None = None) -> "Reads":
        """HBM-resident reads [first_read, first_read + n_reads) of a synth.SynthSpec."""
        n = spec.n_reads if n_reads is None else n_reads
        slen = np.asarray(spec.species_len, dtype=np.uint64)
        thr = np.ascontiguousarray(spec.weight_thresholds(), dtype=np.uint64)
        h = C.c_void_p()
        self.check(lib().mdbg_reads_synthetic(self.h, spec.seed, n, spec.read_len, first_read, _ptr(slen), _ptr(thr),
                                              len(slen), spec.sub_threshold(), spec.ins_threshold(), spec.del_threshold(), spec.window(),
                                              int(spec.with_quality), C.byref(h)))
        return Reads(self, h)

    # -- minimizer space ----------------------------------------------------------------------
    def minimizers_from_host(self, mins: np.ndarray, offsets: np.ndarray) -> "Minimizers":
        mins = np.ascontiguousarray(mins, dtype=np.uint32)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        h = C.c_void_p()
        self.check(lib().mdbg_minimizers_from_host(self.h, _ptr(mins), _ptr(offsets), len(offsets) - 1, C.byref(h)))
        return Minimizers(self, h)

    def scan(self, reads: "Reads", K: int = 15, density: float = 0.005, hpc: bool = True,
             min_read_quality: float = 0.0, repetitive=None, apply_read_filters: bool = True,
             quality_window: int = 0, no_end_trim: bool = False, ignore_qualities: bool = False) -> "Minimizers":
        rep = np.ascontiguousarray(repetitive if repetitive is not None else [], dtype=np.uint32)
        p = ScanParams(K, density, int(hpc), min_read_quality, rep.ctypes.data_as(C.POINTER(C.c_uint32)), len(rep),
                       int(apply_read_filters), int(quality_window), int(no_end_trim), int(ignore_qualities))
        h = C.c_void_p()
        self.check(lib().mdbg_scan(self.h, reads.h, C.byref(p), C.byref(h)))
        return Minimizers(self, h)

    def apply_density_threshold(self, m: "Minimizers", density: float) -> "Minimizers":
        h = C.c_void_p()
        self.check(lib().mdbg_apply_density_threshold(self.h, m.h, C.c_float(density), C.byref(h)))
        return Minimizers(self, h)

    def purge_palindromes(self, m: "Minimizers", first_k: int, last_k: int) -> "Minimizers":
        h = C.c_void_p()
        self.check(lib().mdbg_purge_palindromes(self.h, m.h, first_k, last_k, C.byref(h)))
        return Minimizers(self, h)

    def repetitive_minimizers(self, m: "Minimizers", max_out: int = 4096) -> np.ndarray:
        out = np.zeros(max_out, dtype=np.uint32)
        n = C.c_uint32(max_out)
        self.check(lib().mdbg_repetitive_minimizers(self.h, m.h, _ptr(out), C.byref(n)))
        return out[: n.value].copy()

    def census(self) -> "Census":
        h = C.c_void_p()
        self.check(lib().mdbg_census_create(self.h, C.byref(h)))
        return Census(self, h)

    # -- k-min-mer tables -----------------------------------------------------------------------
    def kminmer_count_first(self, m: "Minimizers", k: int = 4, min_abundance: int = 0) -> "Table":
        h = C.c_void_p()
        self.check(lib().mdbg_kminmer_count_first(self.h, m.h, k, min_abundance, C.byref(h)))
        return Table(self, h)

    def prev_from_records(self, records: bytes | np.ndarray) -> "Table":
        raw = records if isinstance(records, (bytes, bytearray)) else np.ascontiguousarray(records).tobytes()
        h = C.c_void_p()
        self.check(lib().mdbg_prev_from_records(self.h, raw, len(raw) // 20, C.byref(h)))
        return Table(self, h)

    def prev_overlay_unitigs(self, prev: "Table", unitigs: "Minimizers", abundance: np.ndarray, k_prev: int) -> None:
        ab = np.ascontiguousarray(abundance, dtype=np.uint32)
        self.check(lib().mdbg_prev_overlay_unitigs(self.h, prev.h, unitigs.h, _ptr(ab), k_prev))

    def kminmer_count_refined(self, reads: "Minimizers", unitigs: "Minimizers | None", k: int, prev: "Table") -> "Table":
        h = C.c_void_p()
        self.check(lib().mdbg_kminmer_count_refined(self.h, reads.h, unitigs.h if unitigs else None, k, prev.h, C.byref(h)))
        return Table(self, h)

    def kminmer_index(self, reads: "Minimizers", unitigs: "Minimizers | None", k: int, prev: "Table") -> "Table":
        h = C.c_void_p()
        self.check(lib().mdbg_kminmer_index(self.h, reads.h, unitigs.h if unitigs else None, k, prev.h, C.byref(h)))
        return Table(self, h)

    def memcpy_device(self, dst_ptr: int, src_ptr: int, nbytes: int) -> None:
        self.check(lib().mdbg_memcpy_device(self.h, C.c_void_p(dst_ptr), C.c_void_p(src_ptr), nbytes))

    def edge_index(self, nodes: "Table") -> tuple["Table", int]:
        """EdgeIndexer: (table of distinct prefix/suffix identities, checksum as the reference logs it)."""
        h = C.c_void_p()
        ck = C.c_uint64()
        self.check(lib().mdbg_edge_index(self.h, nodes.h, C.byref(h), C.byref(ck)))
        return Table(self, h), ck.value

    def unitig_edge_index(self, unitigs: "Minimizers", k: int) -> tuple["Table", int]:
        """UnitigEdgeIndexer: distinct prefix/suffix identities of the first and last k-min-mer of every unitig."""
        h = C.c_void_p()
        ck = C.c_uint64()
        self.check(lib().mdbg_unitig_edge_index(self.h, unitigs.h, k, C.byref(h), C.byref(ck)))
        return Table(self, h), ck.value

    # -- record files handed over as bytes (mdbg_bytes_*) ------------------------------------------
    def bytes_from_host(self, raw: bytes, piece: int = 1 << 22) -> "DeviceBytes":
        """A file's bytes on the device, uploaded in pieces (tests: from ordinary memory, so every piece is a blocking copy)."""
        h = C.c_void_p()
        self.check(lib().mdbg_bytes_create(self.h, len(raw), C.byref(h)))
        b = DeviceBytes(self, h)
        buf = np.frombuffer(raw, dtype=np.uint8)
        last = C.c_uint64(0)
        for at in range(0, len(raw), piece):
            n = min(piece, len(raw) - at)
            self.check(lib().mdbg_bytes_upload_async(self.h, h, at, buf[at:].ctypes.data_as(C.c_void_p), n, C.byref(last)))
        if last.value:
            rc = lib().mdbg_bytes_upload_done(self.h, h, last.value, 1)
            if rc != 1:
                self.check(rc if rc < 0 else -1)
        return b

    def minimizers_from_record_bytes(self, b: "DeviceBytes", offsets: np.ndarray, want_circular: bool = False):
        """read_data_corrected.txt / unitig_data.txt records taken apart on the device (mdbg_minimizers_from_record_bytes)."""
        offs = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offs) - 1
        circ = np.zeros(n, dtype=np.uint8) if want_circular else None
        h = C.c_void_p()
        self.check(lib().mdbg_minimizers_from_record_bytes(self.h, b.h, _ptr(offs), n, _ptr(circ), C.byref(h)))
        m = Minimizers(self, h)
        return (m, circ) if want_circular else m

    def prev_from_record_bytes(self, b: "DeviceBytes", n_records: int) -> "Table":
        h = C.c_void_p()
        self.check(lib().mdbg_prev_from_record_bytes(self.h, b.h, n_records, C.byref(h)))
        return Table(self, h)

    def small_contigs(self, unitigs: "Minimizers", k: int, k_prev: int, prev: "Table") -> np.ndarray:
        """1 per unitig that IndexKminmerFunctor writes to smallContigs_k<k>.bin instead of indexing (k > 8 is the caller's test)."""
        n = unitigs.info()["n_reads"]
        flags = np.zeros(n, dtype=np.uint8)
        self.check(lib().mdbg_small_contigs(self.h, unitigs.h, k, k_prev, prev.h, flags.ctypes.data))
        return flags

    # -- the exchange inside the library (peer copies or RCCL) --------------------------------------
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        rc = lib().mdbg_comm_unique_id(buf)
        if rc:
            raise MdbgError(rc, (lib().mdbg_last_error(None) or b"").decode())
        return buf.raw

    def minimizers_concat(self, parts: list) -> "Minimizers":
        """The reads of the parts, in order, as one set (appended on the device: mdbg_minimizers_concat)."""
        arr = (C.c_void_p * len(parts))(*[getattr(p.h, "value", p.h) for p in parts])
        h = C.c_void_p()
        self.check(lib().mdbg_minimizers_concat(self.h, arr, len(parts), C.byref(h)))
        return Minimizers(self, h)

    def minimizers_slice(self, m: "Minimizers", first_read: int, n_reads: int) -> "Minimizers":
        """Reads [first_read, first_read + n_reads) of m as a set of their own (mdbg_minimizers_slice)."""
        h = C.c_void_p()
        self.check(lib().mdbg_minimizers_slice(self.h, m.h, first_read, n_reads, C.byref(h)))
        return Minimizers(self, h)

    def comm_create(self, unique_id: bytes, rank: int, n_ranks: int, mode: "int | str" = -1) -> "Comm":
        """mode: "peer" | "rccl" | "auto" (or the MDBG_COMM_* value); -1: the environment's MDBG_COMM_MODE, "auto" when unset."""
        h = C.c_void_p()
        m = COMM_MODES[mode] if isinstance(mode, str) else int(mode)
        self.check(lib().mdbg_comm_create_mode(self.h, unique_id, rank, n_ranks, m, C.byref(h)))
        return Comm(h)

    def kminmer_count_first_sharded(self, comm: "Comm", m: "Minimizers", k: int = 4, min_abundance: int = 0) -> "Table":
        h = C.c_void_p()
        self.check(lib().mdbg_kminmer_count_first_sharded(self.h, comm.h, m.h, k, min_abundance, C.byref(h)))
        return Table(self, h)

    # -- sharded first pass (one process per GPU) ---------------------------------------------------
    def shard_from_table(self, local: "Table", n_ranks: int) -> "Shard":
        """Sharded k > firstK: the rows of a local refined / index table grouped by owner rank (reduce -> keep)."""
        h, d_rows = C.c_void_p(), C.c_void_p()
        counts = np.zeros(n_ranks, dtype=np.uint64)
        self.check(lib().mdbg_shard_from_table(self.h, local.h, n_ranks, C.byref(h), C.byref(d_rows), counts.ctypes.data_as(_u64p)))
        return Shard(self, h, local.info()["k"], d_rows.value or 0, counts)

    def shard_begin(self, m: "Minimizers", k: int, n_ranks: int) -> "Shard":
        h, d_rows = C.c_void_p(), C.c_void_p()
        counts = np.zeros(n_ranks, dtype=np.uint64)
        self.check(lib().mdbg_shard_begin(self.h, m.h, k, n_ranks, C.byref(h), C.byref(d_rows), counts.ctypes.data_as(_u64p)))
        return Shard(self, h, k, d_rows.value or 0, counts)


class Census:
    """Counts of minimizer values over several batches (mdbg_census_*)."""

    def __init__(self, ctx: "Context", h):
        self.ctx, self.h = ctx, h

    def add(self, m: "Minimizers") -> None:
        self.ctx.check(lib().mdbg_census_add(self.ctx.h, self.h, m.h))

    def top(self, max_out: int = 4096) -> np.ndarray:
        out = np.zeros(max_out, dtype=np.uint32)
        n = C.c_uint32(max_out)
        self.ctx.check(lib().mdbg_census_top(self.ctx.h, self.h, _ptr(out), C.byref(n)))
        return out[: n.value].copy()

    def free(self) -> None:
        if self.h:
            lib().mdbg_census_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


COMM_MODES = {"default": -1, "rccl": 0, "peer": 1, "auto": 2}


class Comm:
    """A communicator owned by the library (mdbg_comm_create_mode): peer copies or RCCL."""

    def __init__(self, h):
        self.h = h

    @property
    def mode(self) -> str:
        """The transport it ended up with: "peer" or "rccl" (mdbg_comm_mode)."""
        return {0: "rccl", 1: "peer"}.get(int(lib().mdbg_comm_mode(self.h)), "?")

    @property
    def note(self) -> str:
        """Why "auto" did not take the peer copies ("" if it did): mdbg_comm_note."""
        return (lib().mdbg_comm_note(self.h) or b"").decode()

    def stats(self) -> dict:
        """What the communicator carried so far (mdbg_comm_stats)."""
        st = (C.c_uint64 * 8)()
        ms = C.c_double()
        lib().mdbg_comm_stats(self.h, st, C.byref(ms))
        return dict(rank=int(st[0]), n_ranks=int(st[1]), rccl_ranks=int(st[2]), exchanges=int(st[3]), bytes_to_peers=int(st[4]),
                    bytes_from_peers=int(st[5]), bytes_local=int(st[6]), exchange_ms=float(ms.value), mode=self.mode, **self.times())

    def times(self) -> dict:
        """Where the time inside the exchanges went (mdbg_comm_times): the owner's reduction, waiting, and what is left: the transport's host time."""
        t = (C.c_double * 3)()
        lib().mdbg_comm_times(self.h, t)
        return dict(reduce_ms=float(t[1]), wait_ms=float(t[2]), host_ms=max(0.0, float(t[0]) - float(t[1]) - float(t[2])))

    def abort(self, ctx: "Context", code: int = -1) -> None:
        """This rank cannot enter the exchange its peers are about to enter: tell them (mdbg_shard_abort)."""
        lib().mdbg_shard_abort(ctx.h, self.h, code)

    def destroy(self) -> None:
        if self.h:
            lib().mdbg_comm_destroy(self.h)
            self.h = None


class DeviceView:
    """Zero-copy window on library-owned device memory for torch.as_tensor (__cuda_array_interface__)."""

    def __init__(self, ptr: int, shape: tuple, typestr: str = "<i8"):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (ptr, False), "version": 2}


class Shard:
    """State of one rank's sharded first pass: begin (rows to send) -> reduce (reply) -> finish (table)."""

    def __init__(self, ctx: Context, h, k: int, d_rows: int, counts: np.ndarray):
        self.ctx, self.h, self.k = ctx, h, k
        self.d_rows, self.counts = d_rows, counts
        self.row_words = lib().mdbg_row_words(k)

    @property
    def n_rows(self) -> int:
        return int(self.counts.sum())

    def reduce(self, d_recv: int, n_recv: int) -> int:
        """Device pointer of n_recv u64 replies (global count | bit 63 = lister) aligned with the received rows."""
        d_reply = C.c_void_p()
        self.ctx.check(lib().mdbg_shard_reduce(self.ctx.h, self.h, C.c_void_p(d_recv), n_recv, C.byref(d_reply)))
        return d_reply.value or 0

    def exchange(self, comm: "Comm") -> int:
        """Rows to their owners, reduce, replies back (inside the library, over the communicator's transport: peer copies or RCCL): device pointer of the replies for finish()."""
        d = C.c_void_p()
        self.ctx.check(lib().mdbg_shard_exchange(self.ctx.h, comm.h, self.h, C.c_void_p(self.d_rows), self.counts.ctypes.data_as(_u64p), C.byref(d)))
        return d.value or 0

    def finish(self, d_replies: int, min_abundance: int) -> "Table":
        h = C.c_void_p()
        self.ctx.check(lib().mdbg_shard_finish(self.ctx.h, self.h, C.c_void_p(d_replies), min_abundance, C.byref(h)))
        return Table(self.ctx, h)

    def keep(self, d_replies: int) -> "Table":
        """Sharded k > firstK: the rows of the local table this rank was told to list."""
        h = C.c_void_p()
        self.ctx.check(lib().mdbg_shard_keep(self.ctx.h, self.h, C.c_void_p(d_replies), C.byref(h)))
        return Table(self.ctx, h)

    def free(self):
        if self.h:
            lib().mdbg_shard_free(self.h)
            self.h = None


def exchange_local(ctx: Context, shards: list) -> list:
    """Both exchanges of a sharded pass among shards on one device (mdbg_shard_exchange_local): the replies' device pointers, one per shard."""
    n = len(shards)
    hs = (C.c_void_p * n)(*[s.h for s in shards])
    rows = (C.c_void_p * n)(*[C.c_void_p(s.d_rows) for s in shards])
    counts = np.ascontiguousarray(np.stack([np.asarray(s.counts, dtype=np.uint64) for s in shards]).reshape(-1))
    out = (C.c_void_p * n)()
    ctx.check(lib().mdbg_shard_exchange_local(ctx.h, hs, n, rows, counts.ctypes.data_as(_u64p), out))
    return [o or 0 for o in out]


class Reads:
    def __init__(self, ctx: Context, h):
        self.ctx, self.h = ctx, h

    def info(self) -> dict:
        n, nb, nw = C.c_uint32(), C.c_uint64(), C.c_uint64()
        lib().mdbg_reads_info(self.h, C.byref(n), C.byref(nb), C.byref(nw))
        return dict(n_reads=n.value, n_bases=nb.value, n_words=nw.value)

    def get(self, index: int, with_quality: bool = False):
        L = C.c_uint32()
        self.ctx.check(lib().mdbg_reads_get(self.ctx.h, self.h, index, None, None, C.byref(L)))
        b = C.create_string_buffer(L.value + 1)
        q = C.create_string_buffer(L.value + 1) if with_quality else None
        self.ctx.check(lib().mdbg_reads_get(self.ctx.h, self.h, index, b, q, C.byref(L)))
        return (b.raw[: L.value], q.raw[: L.value]) if with_quality else b.raw[: L.value]

    def export_ascii(self, first: int, count: int) -> tuple[np.ndarray, np.ndarray]:
        """(bases u8[], offsets u64[count+1]) of reads [first, first+count)."""
        nb = C.c_uint64()
        self.ctx.check(lib().mdbg_reads_export_ascii(self.ctx.h, self.h, first, count, None, None, C.byref(nb)))
        bases = np.zeros(nb.value, dtype=np.uint8)
        offs = np.zeros(count + 1, dtype=np.uint64)
        self.ctx.check(lib().mdbg_reads_export_ascii(self.ctx.h, self.h, first, count, _ptr(bases), _ptr(offs), C.byref(nb)))
        return bases, offs

    def mark_ascii(self, index: list[int], seqs: list[bytes]) -> None:
        """The listed reads of a packed batch again as characters: their side masks are derived on the device (mdbg_reads_mark_ascii)."""
        idx = np.ascontiguousarray(index, dtype=np.uint32)
        offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
        np.cumsum([len(s) for s in seqs], out=offs[1:])
        self.ctx.check(lib().mdbg_reads_mark_ascii(self.ctx.h, self.h, _ptr(idx), len(idx), b"".join(seqs), _ptr(offs)))

    def wait(self) -> None:
        """An asynchronous upload has arrived (mdbg_reads_wait)."""
        self.ctx.check(lib().mdbg_reads_wait(self.ctx.h, self.h))

    def export_qualities(self, first: int, count: int) -> np.ndarray:
        """phred+33 bytes of reads [first, first+count), concatenated (offsets as export_ascii's)."""
        nb = C.c_uint64()
        self.ctx.check(lib().mdbg_reads_export_qualities(self.ctx.h, self.h, first, count, None, C.byref(nb)))
        q = np.zeros(nb.value, dtype=np.uint8)
        self.ctx.check(lib().mdbg_reads_export_qualities(self.ctx.h, self.h, first, count, _ptr(q), C.byref(nb)))
        return q

    def free(self) -> None:
        if self.h:
            lib().mdbg_reads_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceBytes:
    def __init__(self, ctx: "Context", h):
        self.ctx, self.h = ctx, h

    def free(self) -> None:
        if self.h:
            lib().mdbg_bytes_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Minimizers:
    def __init__(self, ctx: Context, h):
        self.ctx, self.h = ctx, h

    def info(self) -> dict:
        n, t = C.c_uint32(), C.c_uint64()
        lib().mdbg_minimizers_info(self.h, C.byref(n), C.byref(t))
        return dict(n_reads=n.value, n_minimizers=t.value)

    def to_host(self, full: bool = True) -> dict:
        i = self.info()
        n, t = i["n_reads"], i["n_minimizers"]
        out = dict(offsets=np.zeros(n + 1, np.uint64), minimizers=np.zeros(t, np.uint32))
        if full:
            out.update(pos=np.zeros(t, np.uint32), dir=np.zeros(t, np.uint8), qual=np.zeros(t, np.uint8),
                       read_length=np.zeros(n, np.uint32), mean_quality=np.zeros(n, np.float32),
                       flags=np.zeros(n, np.uint8))
        self.ctx.check(lib().mdbg_minimizers_to_host(
            self.ctx.h, self.h, _ptr(out["offsets"]), _ptr(out["minimizers"]), _ptr(out.get("pos")), _ptr(out.get("dir")),
            _ptr(out.get("qual")), _ptr(out.get("read_length")), _ptr(out.get("mean_quality")), _ptr(out.get("flags"))))
        return out

    def device_ptrs(self) -> tuple[int, int]:
        a, b = C.c_void_p(), C.c_void_p()
        lib().mdbg_minimizers_device_ptrs(self.h, C.byref(a), C.byref(b))
        return a.value or 0, b.value or 0

    def free(self) -> None:
        if self.h:
            lib().mdbg_minimizers_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Table:
    def __init__(self, ctx: Context, h):
        self.ctx, self.h = ctx, h

    def info(self) -> dict:
        k, n, ns, hv = C.c_uint32(), C.c_uint64(), C.c_uint64(), C.c_int()
        lib().mdbg_table_info(self.h, C.byref(k), C.byref(n), C.byref(ns), C.byref(hv))
        return dict(k=k.value, n_records=n.value, n_solid=ns.value, has_vectors=bool(hv.value))

    def to_host(self) -> tuple[np.ndarray, np.ndarray | None]:
        """(records as formats.ABUNDANCE_DTYPE array, vectors u32[n,k] or None)."""
        from .formats import ABUNDANCE_DTYPE
        i = self.info()
        rec = np.zeros(i["n_records"], dtype=ABUNDANCE_DTYPE)
        vec = np.zeros((i["n_records"], i["k"]), dtype=np.uint32) if i["has_vectors"] else None
        self.ctx.check(lib().mdbg_table_to_host(self.ctx.h, self.h, _ptr(rec), _ptr(vec)))
        return rec, vec

    def stats(self) -> dict:
        """What the pass that built the table walked (mdbg_table_stats)."""
        s = (C.c_uint64 * 4)()
        lib().mdbg_table_stats(self.h, s)
        return dict(minimizers=int(s[0]), instances=int(s[1]), keys=int(s[2]), slots=int(s[3]))

    def checksum(self) -> tuple:
        """(sum abundance * hash_lo -- the reference's "Checksum kminmer abundance" --, sum abundance, sum hash_hi, vector sum),
        all modulo 2^64, computed on the device (mdbg_table_checksum)."""
        s = (C.c_uint64 * 4)()
        self.ctx.check(lib().mdbg_table_checksum(self.ctx.h, self.h, s))
        return tuple(int(x) for x in s)

    def to_host_range(self, first: int, count: int) -> tuple[np.ndarray, np.ndarray | None]:
        """Rows [first, first + count) as to_host gives them."""
        from .formats import ABUNDANCE_DTYPE
        i = self.info()
        rec = np.zeros(count, dtype=ABUNDANCE_DTYPE)
        vec = np.zeros((count, i["k"]), dtype=np.uint32) if i["has_vectors"] else None
        self.ctx.check(lib().mdbg_table_to_host_range(self.ctx.h, self.h, first, count, _ptr(rec), _ptr(vec)))
        return rec, vec

    def keys_to_host(self) -> np.ndarray:
        """(n, 2) u64 array of (lo, hi) -- the 16-byte little-endian u128 records of edges.bin."""
        n = self.info()["n_records"]
        out = np.zeros((n, 2), dtype=np.uint64)
        self.ctx.check(lib().mdbg_table_keys_to_host(self.ctx.h, self.h, _ptr(out)))
        return out

    def lookup(self, lo: np.ndarray, hi: np.ndarray) -> np.ndarray:
        lo = np.ascontiguousarray(lo, dtype=np.uint64)
        hi = np.ascontiguousarray(hi, dtype=np.uint64)
        out = np.zeros(len(lo), dtype=np.uint32)
        self.ctx.check(lib().mdbg_table_lookup(self.ctx.h, self.h, _ptr(lo), _ptr(hi), len(lo), _ptr(out)))
        return out

    def free(self) -> None:
        if self.h:
            lib().mdbg_table_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
