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
    (readSelection/ReadSelection.hpp:415-467)."""
    offs = h["offsets"]
    parts = []
    mq = np.asarray(h["mean_quality"], dtype="<f4")
    for r in range(len(offs) - 1):
        a, b = int(offs[r]), int(offs[r + 1])
        parts.append(struct.pack("<IB", b - a, 0))
        parts.append(np.asarray(h["minimizers"][a:b], "<u4").tobytes())
        parts.append(np.asarray(h["pos"][a:b], "<u4").tobytes())
        parts.append(np.asarray(h["dir"][a:b], "u1").tobytes())
        parts.append(np.asarray(h["qual"][a:b], "u1").tobytes())
        parts.append(mq[r:r + 1].tobytes())
        parts.append(struct.pack("<I", int(h["read_length"][r])))
    return b"".join(parts)
