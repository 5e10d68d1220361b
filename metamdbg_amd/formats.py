"""On-disk formats of the reference's minimizer / k-min-mer hot path (little-endian, packed).

These are the drop-in contract of the two sub-commands; citations are
file:line under /root/reference/src.

* ``parameters.gz``            pipeline/AssemblyPipeline.hpp:1479-1517 (write), Commons.hpp:1475-1497 (read)
* ``read_stats.txt``           readSelection/ReadSelection.hpp:372-378
* ``read_data_init.txt``       readSelection/ReadSelection.hpp:415-467
* ``read_data_corrected.txt``  readSelection/ReadSelection.hpp:1420-1426
* ``kminmerData_min.txt``      Commons.hpp:4429-4446  (u32[k] per record)
* ``kminmerData_abundance.txt``Commons.hpp:4463-4472  (u128 hash little-endian = lo,hi ; u32 abundance)
"""
from __future__ import annotations

import gzip
import struct
from dataclasses import dataclass

import numpy as np

PARAMS_STRUCT = "<QQfQfffQQQf?iQ"  # 14 raw fields, no padding (gzwrite field by field)


@dataclass
class Parameters:
    minimizer_size: int = 15
    kminmer_size: int = 4
    density: float = 0.005
    first_k: int = 4
    prev_k: int = 4
    last_k: int = 0
    mean_read_length: int = 0
    correction_density: float = 0.025
    hpc: bool = True
    data_type: int = 0  # 0 HiFi, 1 ONT
    snpmer_size: int = 0

    def pack(self) -> bytes:
        d = np.float32(self.density)
        spacing = np.float32(1) / d                       # AssemblyPipeline.hpp:1486-1488 (float math)
        klen = np.float32(spacing * np.float32(self.kminmer_size - 1))
        kovl = np.float32(klen - spacing)
        return struct.pack(
            PARAMS_STRUCT, self.minimizer_size, self.kminmer_size, float(d), self.first_k,
            float(spacing), float(klen), float(kovl), self.prev_k & 0xFFFFFFFFFFFFFFFF, self.last_k,
            self.mean_read_length, float(np.float32(self.correction_density)), self.hpc,
            self.data_type, self.snpmer_size)

    def save(self, path: str) -> None:
        with gzip.open(path, "wb") as f:
            f.write(self.pack())

    @staticmethod
    def load(path: str) -> "Parameters":
        with gzip.open(path, "rb") as f:
            raw = f.read()
        v = struct.unpack(PARAMS_STRUCT, raw[: struct.calcsize(PARAMS_STRUCT)])
        return Parameters(minimizer_size=v[0], kminmer_size=v[1], density=v[2], first_k=v[3],
                          prev_k=v[7], last_k=v[8], mean_read_length=v[9], correction_density=v[10],
                          hpc=v[11], data_type=v[12], snpmer_size=v[13])


READ_STATS_STRUCT = "<QIfQfIQ"  # nReads, N50, density, bases, avgQ, meanLen, nMinimizers (40 B)


def parse_read_stats(raw: bytes) -> dict:
    v = struct.unpack(READ_STATS_STRUCT, raw[:40])
    return dict(n_reads=v[0], n50=v[1], density=v[2], n_bases=v[3], avg_quality=v[4],
                mean_length=v[5], n_minimizers=v[6])


def parse_read_data_init(raw: bytes) -> list[dict]:
    """u32 n; u8 circ; u32 m[n]; u32 pos[n]; u8 dir[n]; u8 qual[n]; f32 meanQ; u32 readLen."""
    out, o = [], 0
    while o < len(raw):
        n, circ = struct.unpack_from("<IB", raw, o)
        o += 5
        m = np.frombuffer(raw, "<u4", n, o); o += 4 * n
        pos = np.frombuffer(raw, "<u4", n, o); o += 4 * n
        d = np.frombuffer(raw, "u1", n, o); o += n
        q = np.frombuffer(raw, "u1", n, o); o += n
        meanq_bits, rl = struct.unpack_from("<II", raw, o); o += 8
        out.append(dict(minimizers=m, pos=pos, dir=d, qual=q, mean_quality_bits=meanq_bits,
                        read_length=rl, circular=circ))
    return out


def parse_minimizer_reads(raw: bytes) -> tuple[np.ndarray, np.ndarray]:
    """``read_data_corrected.txt`` / ``unitig_data.txt``: u32 n; u8 circ; u32 m[n].
    Returns CSR (minimizers u32, offsets u64[n_reads+1])."""
    mins, offs, o = [], [0], 0
    while o < len(raw):
        (n,) = struct.unpack_from("<I", raw, o)
        o += 5
        mins.append(np.frombuffer(raw, "<u4", n, o)); o += 4 * n
        offs.append(offs[-1] + n)
    m = np.concatenate(mins) if mins else np.zeros(0, "<u4")
    return np.ascontiguousarray(m, dtype=np.uint32), np.asarray(offs, dtype=np.uint64)


def write_minimizer_reads(mins: np.ndarray, offs: np.ndarray) -> bytes:
    parts = []
    for r in range(len(offs) - 1):
        a, b = int(offs[r]), int(offs[r + 1])
        parts.append(struct.pack("<IB", b - a, 0))
        parts.append(np.asarray(mins[a:b], "<u4").tobytes())
    return b"".join(parts)


ABUNDANCE_DTYPE = np.dtype([("lo", "<u8"), ("hi", "<u8"), ("abundance", "<u4")])  # 20 B packed


def parse_unitig_nodes(raw: bytes) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
    """unitigGraph.nodes.bin (graph/CreateMdbg.hpp:4335-4343): records `u32 n; u32 minimizers[n]; u32 unitigIndex`.
    Returns (minimizers u32[], offsets u64[n_unitigs+1], unitigIndex u32[])."""
    a = np.frombuffer(raw, dtype="<u4")
    mins, offs, idx = [], [0], []
    i = 0
    while i < len(a):
        n = int(a[i])
        mins.append(a[i + 1: i + 1 + n])
        idx.append(int(a[i + 1 + n]))
        offs.append(offs[-1] + n)
        i += n + 2
    return (np.concatenate(mins).astype(np.uint32) if mins else np.zeros(0, np.uint32),
            np.asarray(offs, dtype=np.uint64), np.asarray(idx, dtype=np.uint32))


def parse_abundance_table(raw: bytes) -> np.ndarray:
    return np.frombuffer(raw, ABUNDANCE_DTYPE)


def sorted_abundance_records(raw_or_arr) -> np.ndarray:
    """Canonical multiset form of ``kminmerData_abundance.txt``: records sorted by (hi, lo, abundance)."""
    a = parse_abundance_table(raw_or_arr) if isinstance(raw_or_arr, (bytes, bytearray)) else raw_or_arr
    order = np.lexsort((a["abundance"], a["lo"], a["hi"]))
    return a[order]


def sorted_vector_records(raw: bytes, k: int) -> np.ndarray:
    """Canonical multiset form of ``kminmerData_min.txt``: rows of k u32 sorted lexicographically."""
    v = np.frombuffer(raw, "<u4").reshape(-1, k)
    order = np.lexsort(tuple(v[:, c] for c in range(k - 1, -1, -1)))
    return v[order]


def build_read_data_init(h: dict) -> bytes:
    """Serialise mdbg_scan output (capi.Minimizers.to_host(full=True)) as ``read_data_init.txt``
    (readSelection/ReadSelection.hpp:415-467): u32 n; u8 circ; u32 m[n]; u32 pos[n]; u8 dir[n]; u8 qual[n]; f32 meanQ; u32 len."""
    offs = np.asarray(h["offsets"], dtype=np.int64)
    n_reads = len(offs) - 1
    cnt = np.diff(offs)
    total = int(offs[-1])
    start = 13 * np.arange(n_reads, dtype=np.int64) + 10 * offs[:-1]            # first byte of every record
    out = np.zeros(13 * n_reads + 10 * total, dtype=np.uint8)

    def put_u32(at, v):
        v = np.asarray(v).astype(np.uint32)
        for b in range(4):
            out[at + b] = ((v >> np.uint32(8 * b)) & np.uint32(255)).astype(np.uint8)

    put_u32(start, cnt)                                                          # circ byte stays 0
    if total:
        rd = np.repeat(np.arange(n_reads, dtype=np.int64), cnt)                  # read of every minimizer
        j = np.arange(total, dtype=np.int64) - offs[rd]                          # its index inside the read
        base, n_of = start[rd] + 5, cnt[rd]
        put_u32(base + 4 * j, h["minimizers"])
        put_u32(base + 4 * n_of + 4 * j, h["pos"])
        out[base + 8 * n_of + j] = np.asarray(h["dir"], dtype=np.uint8)
        out[base + 9 * n_of + j] = np.asarray(h["qual"], dtype=np.uint8)
    tail = start + 5 + 10 * cnt
    put_u32(tail, np.asarray(h["mean_quality"], dtype="<f4").view("<u4"))
    put_u32(tail + 4, h["read_length"])
    return out.tobytes()


# ---- order-independent digests (tests/golden/hifi_1m: tables of a million reads are compared through these) ----
def _sha(a: np.ndarray) -> str:
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def minimizer_reads_digest(mins: np.ndarray, offs: np.ndarray) -> str:
    """Digest of a multiset of minimizer-space reads (``read_data_corrected.txt``: the reference writes its records in thread
    order): one 64-bit hash per read over (length, values in order), the hashes sorted, sha256 of that."""
    return _sha(np.sort(_read_hashes(mins, offs)))


def _read_hashes(mins: np.ndarray, offs: np.ndarray) -> np.ndarray:
    offs = np.asarray(offs, dtype=np.int64)
    cnt = np.diff(offs)
    m = np.asarray(mins).astype(np.uint64)
    with np.errstate(over="ignore"):
        h = cnt.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        if len(m):
            rd = np.repeat(np.arange(len(cnt), dtype=np.int64), cnt)
            j = (np.arange(len(m), dtype=np.int64) - offs[rd]).astype(np.uint64)
            w = (j + np.uint64(1)) * np.uint64(0xD6E8FEB86659FD93)
            w ^= w >> np.uint64(32)
            w = w * np.uint64(0xD6E8FEB86659FD93) | np.uint64(1)
            np.add.at(h, rd, (m + np.uint64(0x632BE59BD9B4E019)) * w)
    return h


def minimizer_reads_equal_as_multisets(m1, o1, m2, o2) -> bool:
    """Exact comparison of two sets of minimizer-space reads as multisets of reads (the reference writes
    read_data_corrected.txt in thread order): both are put in the order of a per-read hash and compared value for value."""
    o1, o2 = np.asarray(o1, dtype=np.int64), np.asarray(o2, dtype=np.int64)
    if len(o1) != len(o2) or int(o1[-1]) != int(o2[-1]):
        return False

    def ordered(m, o):
        h = _read_hashes(m, o)
        order = np.argsort(h, kind="stable")
        cnt = np.diff(o)[order]
        start = np.concatenate([[0], np.cumsum(cnt)])[:-1]
        idx = np.repeat(o[:-1][order] - start, cnt) + np.arange(int(cnt.sum()), dtype=np.int64)
        return h[order], cnt, np.asarray(m)[idx]
    h1, c1, v1 = ordered(m1, o1)
    h2, c2, v2 = ordered(m2, o2)
    return bool(np.array_equal(h1, h2) and np.array_equal(c1, c2) and np.array_equal(v1, v2))


def table_digests(records, vectors, k: int) -> dict:
    """sha256 of the canonical (sorted) forms of ``kminmerData_abundance.txt`` / ``kminmerData_min.txt``."""
    out = {"abundance_sorted_sha256": _sha(sorted_abundance_records(records))}
    if vectors is not None:
        raw = vectors if isinstance(vectors, (bytes, bytearray)) else np.asarray(vectors, dtype="<u4").tobytes()
        out["min_sorted_sha256"] = _sha(sorted_vector_records(raw, k))
    return out
