"""Seeded synthetic read sets, reproducible on host and device.

Everything is a pure function of (seed, read index, base index) through the splitmix64
finaliser, so the device generator in ``csrc/synth.hip`` (used by bench.py to build HBM-resident
inputs without a PCIe copy) and this numpy version emit identical reads; tests check that.

Base codes follow the reference's 2-bit code ``(c >> 1) & 3`` (utils/kmer/Kmer.hpp:462):
A=0, C=1, T=2, G=3; complement is ``code ^ 2``.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

CODE2ASCII = np.frombuffer(b"ACTG", dtype=np.uint8)
M64 = np.uint64(0xFFFFFFFFFFFFFFFF)

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_C1 = np.uint64(0xBF58476D1CE4E5B9)
_C2 = np.uint64(0x94D049BB133111EB)


def mix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on u64 arrays (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        z = (np.asarray(x, dtype=np.uint64) + _GOLD)
        z = (z ^ (z >> np.uint64(30))) * _C1
        z = (z ^ (z >> np.uint64(27))) * _C2
        return z ^ (z >> np.uint64(31))


@dataclass
class SynthSpec:
    """A metagenome-like read set: ``species`` random genomes with relative abundances."""
    n_reads: int
    read_len: int = 10_000
    seed: int = 42
    sub_rate: float = 0.001            # HiFi: substitution-only 0.1 %
    ins_rate: float = 0.0              # ONT R10: 1 % substitutions + 0.5 % insertions + 0.5 % deletions
    del_rate: float = 0.0
    species_len: list[int] = field(default_factory=lambda: [2_000_000])
    species_weight: list[float] = field(default_factory=lambda: [1.0])
    with_quality: bool = False         # FASTQ with phred uniform 10..39
    name: str = "hifi"

    def genome_offsets(self) -> np.ndarray:
        return np.concatenate([[0], np.cumsum(self.species_len)]).astype(np.uint64)

    def weight_thresholds(self) -> np.ndarray:
        """u64 cumulative thresholds: species s is chosen when u < thr[s] (first match)."""
        w = np.asarray(self.species_weight, dtype=np.float64)
        c = np.cumsum(w) / w.sum()
        thr = np.minimum(np.floor(c * 2.0**64), 2.0**64 - 2048).astype(np.uint64)
        thr[-1] = M64
        return thr

    def sub_threshold(self) -> int:
        return min(int(self.sub_rate * 2.0**64), 2**64 - 1)

    def ins_threshold(self) -> int:
        return min(int(self.ins_rate * 2.0**64), 2**62)

    def del_threshold(self) -> int:
        return min(int(self.del_rate * 2.0**64), 2**62)

    def window(self) -> int:
        """Genome bases set aside for one read: its length, plus room for what deletions consume (reads with
        indels; several standard deviations above the expected number)."""
        if not (self.ins_rate or self.del_rate):
            return self.read_len
        return self.read_len + int(3.0 * self.del_rate * self.read_len) + 64


def hifi_spec(n_reads: int, seed: int = 42, read_len: int = 10_000, coverage: float = 50.0) -> SynthSpec:
    """~`coverage`x total, several species with spread abundances."""
    total = max(int(n_reads * read_len / coverage), 4 * read_len)
    fr = np.array([0.4, 0.3, 0.2, 0.1])            # genome length share
    wt = np.array([0.2, 0.3, 0.25, 0.25])          # read share -> coverage 25x..125x
    lens = np.maximum((fr * total).astype(np.int64), 2 * read_len)
    return SynthSpec(n_reads=n_reads, read_len=read_len, seed=seed, sub_rate=0.001,
                     species_len=[int(x) for x in lens], species_weight=[float(x) for x in wt], name="hifi")


def ont_spec(n_reads: int, seed: int = 42, read_len: int = 20_000, coverage: float = 50.0) -> SynthSpec:
    """SURVEY 8(d) ONT R10: 20 kb reads, 2 % errors (1 % substitutions, 0.5 % insertions, 0.5 % deletions), FASTQ with
    phred uniform 10..39, the same species structure as hifi_spec."""
    h = hifi_spec(n_reads, seed=seed, read_len=read_len, coverage=coverage)
    spec = SynthSpec(n_reads=n_reads, read_len=read_len, seed=seed, sub_rate=0.01, ins_rate=0.005, del_rate=0.005,
                     species_len=h.species_len, species_weight=h.species_weight, with_quality=True, name="ont")
    spec.species_len = [max(int(x), 2 * spec.window()) for x in spec.species_len]
    return spec


def genome_codes(spec: SynthSpec, start: int = 0, stop: int | None = None) -> np.ndarray:
    """2-bit codes of the concatenated genomes, positions [start, stop)."""
    g = int(spec.genome_offsets()[-1])
    stop = g if stop is None else stop
    idx = np.arange(start, stop, dtype=np.uint64)
    with np.errstate(over="ignore"):
        key = mix64(np.uint64(spec.seed)) + idx
    return (mix64(key) >> np.uint64(62)).astype(np.uint8)


def read_layout(spec: SynthSpec, r0: int, r1: int):
    """(genome start, strand) of reads [r0, r1)."""
    r = np.arange(r0, r1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        base = mix64(np.uint64(spec.seed) ^ np.uint64(0xA5A5A5A5A5A5A5A5))
        u_species = mix64(base + np.uint64(4) * r)
        u_start = mix64(base + np.uint64(4) * r + np.uint64(1))
        u_strand = mix64(base + np.uint64(4) * r + np.uint64(2))
    thr = spec.weight_thresholds()
    sp = np.searchsorted(thr, u_species, side="right")  # first s with u < thr[s]
    sp = np.minimum(sp, len(thr) - 1)
    offs = spec.genome_offsets()
    span = (np.asarray(spec.species_len, dtype=np.uint64)[sp] - np.uint64(spec.window()) + np.uint64(1))
    start = offs[sp] + (u_start % span)
    strand = (u_strand & np.uint64(1)).astype(np.uint8)
    return start, strand


def read_codes(spec: SynthSpec, r0: int, r1: int, genome: np.ndarray | None = None) -> np.ndarray:
    """2-bit codes, shape (r1-r0, read_len), of reads [r0, r1) including sequencing errors.

    One draw e = mix(mix(seed3 + r) + i) per READ position i decides what happens there, in this order of the u64 range:
    e < T_ins: an inserted base (consumes no genome); then T_del: one genome base is skipped before this one; then T_sub:
    the genome base is replaced by one of the other three.  Position i of a read therefore shows genome base
    k(i) = i - #insertions before i + #deletions up to and including i of its window (counted from the window's end and
    complemented on the reverse strand).  Without indels k(i) = i and the window is the read."""
    if genome is None:
        genome = genome_codes(spec)
    L, W = spec.read_len, spec.window()
    start, strand = read_layout(spec, r0, r1)
    r = np.arange(r0, r1, dtype=np.uint64)[:, None]
    with np.errstate(over="ignore"):
        rk = mix64(np.uint64(spec.seed) ^ np.uint64(0x5EED5EED5EED5EED)) + r
        e = mix64(mix64(rk) + np.arange(L, dtype=np.uint64)[None, :])
    t_ins, t_del, t_sub = spec.ins_threshold(), spec.del_threshold(), spec.sub_threshold()
    ins = e < np.uint64(t_ins)
    dele = ~ins & (e < np.uint64(t_ins + t_del))
    hit = ~ins & ~dele & (e < np.uint64(min(t_ins + t_del + t_sub, 2**64 - 1)))
    k = np.arange(L, dtype=np.int64)[None, :] - (np.cumsum(ins, axis=1) - ins) + np.cumsum(dele, axis=1)
    k = np.minimum(k, W - 1)                       # never reached: the window has room for > 3x the expected deletions
    s = start.astype(np.int64)[:, None]
    fwd = genome[s + k]
    rev = genome[s + (W - 1) - k] ^ np.uint8(2)
    codes = np.where(strand[:, None] == 0, fwd, rev).astype(np.uint8)
    if hit.any():
        delta = (mix64(e[hit]) % np.uint64(3)).astype(np.uint8) + np.uint8(1)
        codes[hit] = (codes[hit] + delta) & np.uint8(3)
    if ins.any():
        codes[ins] = (mix64(e[ins] ^ np.uint64(0x1B5E47ED)) >> np.uint64(62)).astype(np.uint8)
    return codes


def read_qualities(spec: SynthSpec, r0: int, r1: int) -> np.ndarray:
    """ASCII phred+33 qualities (uniform 10..39), shape (r1-r0, read_len)."""
    r = np.arange(r0, r1, dtype=np.uint64)[:, None]
    with np.errstate(over="ignore"):
        rk = mix64(np.uint64(spec.seed) ^ np.uint64(0x0123456789ABCDEF)) + r
        q = mix64(mix64(rk) + np.arange(spec.read_len, dtype=np.uint64)[None, :])
    return ((q % np.uint64(30)) + np.uint64(10 + 33)).astype(np.uint8)


def codes_to_ascii(codes: np.ndarray) -> np.ndarray:
    return CODE2ASCII[codes]


def ascii_to_codes(seq: bytes | np.ndarray) -> np.ndarray:
    a = np.frombuffer(seq, dtype=np.uint8) if isinstance(seq, (bytes, bytearray)) else seq
    return ((a >> 1) & 3).astype(np.uint8)


def write_fasta(path: str, spec: SynthSpec, chunk: int = 2000) -> None:
    genome = genome_codes(spec)
    with open(path, "wb") as f:
        for r0 in range(0, spec.n_reads, chunk):
            r1 = min(r0 + chunk, spec.n_reads)
            asc = codes_to_ascii(read_codes(spec, r0, r1, genome))
            qual = read_qualities(spec, r0, r1) if spec.with_quality else None
            for j in range(r1 - r0):
                if qual is None:
                    f.write(b">r%d\n" % (r0 + j)); f.write(asc[j].tobytes()); f.write(b"\n")
                else:
                    f.write(b"@r%d\n" % (r0 + j)); f.write(asc[j].tobytes()); f.write(b"\n+\n")
                    f.write(qual[j].tobytes()); f.write(b"\n")


# ---- 2-bit packing (device input layout, DESIGN.md "data layout") -------------------------

WORD_BASES = 32          # one u64 word holds 32 bases, base i at bits [2i, 2i+2) (LSB first)
READ_ALIGN_BASES = 64    # every read starts on a 16-byte boundary


def pack_reads(seqs: list[np.ndarray]) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Pack 2-bit code arrays into the device layout.

    Returns (words u64[], read_word_offset u64[n+1], read_len u32[n]); read r occupies
    words[read_word_offset[r] : read_word_offset[r+1]], padded with zero bases to 64-base units.
    """
    n = len(seqs)
    lens = np.array([len(s) for s in seqs], dtype=np.uint32)
    units = (lens.astype(np.uint64) + np.uint64(READ_ALIGN_BASES - 1)) // np.uint64(READ_ALIGN_BASES)
    woff = np.concatenate([[0], np.cumsum(units * np.uint64(2))]).astype(np.uint64)
    words = np.zeros(int(woff[-1]), dtype=np.uint64)
    shifts = (np.arange(WORD_BASES, dtype=np.uint64) * np.uint64(2))[None, :]
    for r in range(n):
        nb = int(units[r]) * READ_ALIGN_BASES
        buf = np.zeros(nb, dtype=np.uint64)
        buf[: lens[r]] = seqs[r]
        w = (buf.reshape(-1, WORD_BASES) << shifts).sum(axis=1, dtype=np.uint64)
        words[int(woff[r]): int(woff[r + 1])] = w
    return words, woff, lens
