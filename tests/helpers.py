"""Shared test helpers: fixture loading and synthetic read regeneration."""
from __future__ import annotations

import hashlib
import json
import os

import numpy as np

from metamdbg_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_manifest(name: str) -> dict:
    with open(os.path.join(GOLDEN, name, "manifest.json")) as f:
        return json.load(f)


def spec_from_manifest(m: dict) -> synth.SynthSpec:
    return synth.SynthSpec(n_reads=m["n_reads"], read_len=m["read_len"], seed=m["seed"], sub_rate=m["sub_rate"],
                           ins_rate=m.get("ins_rate", 0.0), del_rate=m.get("del_rate", 0.0),
                           species_len=m["species_len"], species_weight=m["species_weight"],
                           with_quality=m.get("with_quality", False), name=m["kind"])


def regenerate_reads(m: dict):
    """(list of ASCII read bytes, list of quality bytes or None) + check the FASTA/FASTQ checksum."""
    spec = spec_from_manifest(m)
    asc = synth.codes_to_ascii(synth.read_codes(spec, 0, spec.n_reads))
    qual = synth.read_qualities(spec, 0, spec.n_reads) if spec.with_quality else None
    h = hashlib.sha256()
    for j in range(spec.n_reads):
        if qual is None:
            h.update(b">r%d\n" % j + asc[j].tobytes() + b"\n")
        else:
            h.update(b"@r%d\n" % j + asc[j].tobytes() + b"\n+\n" + qual[j].tobytes() + b"\n")
    assert h.hexdigest() == m["fasta_sha256"], "synthetic generator drifted from the fixture's input"
    seqs = [asc[j].tobytes() for j in range(spec.n_reads)]
    quals = [qual[j].tobytes() for j in range(spec.n_reads)] if qual is not None else None
    return seqs, quals


def read_fasta(path: str) -> list[bytes]:
    seqs = []
    with open(path, "rb") as f:
        for line in f:
            if not line.startswith(b">"):
                seqs.append(line.rstrip(b"\n"))
    return seqs


def read_fastq(path: str):
    seqs, quals = [], []
    with open(path, "rb") as f:
        lines = f.read().split(b"\n")
    for i in range(0, len(lines) - 1, 4):
        seqs.append(lines[i + 1]); quals.append(lines[i + 3])
    return seqs, quals


def golden_bytes(*parts: str) -> bytes:
    with open(os.path.join(GOLDEN, *parts), "rb") as f:
        return f.read()
