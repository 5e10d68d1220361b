"""The CPU oracle (oracle/mdbg_oracle.c) against the golden vectors produced by the REFERENCE's
own code (tests/golden/make_golden.py).  CPU only."""
from __future__ import annotations

import json
import os

import numpy as np
import pytest

from metamdbg_amd import formats
from oracle import pyoracle as orc
from tests import helpers as H


@pytest.fixture(scope="module")
def fn_golden():
    with open(os.path.join(H.GOLDEN, "fn", "fn_golden.json")) as f:
        return json.load(f)


def test_known_answer_scalars():
    # SURVEY.md section 8(c)
    assert orc.kmer_hash(12345) == 8367407816444978832
    assert orc.kmer_hash(2**64 - 1) == 13161889351522953593
    assert orc.density_threshold(0.005) == 92233718306963448
    assert orc.density_threshold(0.025) == 461168608714686432
    rev, vec, hi, lo = orc.kminmer_normalize_hash([5, 9, 3, 7])
    assert (rev, hi, lo) == (0, 13950843744880946374, 7379975562370095872)
    rev, vec, _, _ = orc.kminmer_normalize_hash([3, 7, 7, 3])
    assert rev == 1 and vec.tolist() == [3, 7, 7, 3]


def test_density_threshold_is_exact_boundary():
    for d in (0.005, 0.025, 0.02, 0.05, 0.001, 0.5):
        T = orc.density_threshold(d)
        bound = float(np.float32(d)) * 2.0**64
        assert float(T) >= bound and float(T - 1) < bound


def test_murmur(fn_golden):
    g = fn_golden["murmur"]
    for v, out in zip(g["inputs"], g["outputs"]):
        assert orc.kmer_hash(int(v)) == int(out)


def test_kminmer_normalize_hash(fn_golden):
    for k, g in fn_golden["kminmer"].items():
        for line, out in zip(g["inputs"], g["outputs"]):
            vec = [int(x) for x in line.split()]
            o = [int(x) for x in out.split()]
            rev, cvec, hi, lo = orc.kminmer_normalize_hash(vec)
            assert [rev, hi, lo] + cvec.tolist() == o, (k, line)


def test_purge_palindrome(fn_golden):
    for key, g in fn_golden["purge"].items():
        fk, lk = map(int, key.split("_"))
        for line, out in zip(g["inputs"], g["outputs"]):
            got = orc.purge_palindrome([int(x) for x in line.split()], fk, lk).tolist()
            assert got == [int(x) for x in out.split()], (key, line)


def test_last_k(fn_golden):
    for c in fn_golden["lastk"]:
        d, n50, fk, mk = c["args"]
        assert orc.lib().orc_compute_last_k(d, n50, fk, mk) == c["out"]


@pytest.mark.parametrize("section", ["scan_n", "scan_case"])
def test_scan_with_invalid_characters(fn_golden, section):
    """scan_n: N / n inside reads.  scan_case: mixed case, soft-masked blocks and IUPAC letters -- EncoderRLE compares
    characters (Commons.hpp:4177-4178), the k-mer model sees 2-bit codes (utils/kmer/Kmer.hpp:462)."""
    for key, g in fn_golden[section].items():
        for seq, out in zip(g["inputs"], g["outputs"]):
            toks = out.split()
            exp = [tuple(int(x) for x in t.split(":")) for t in toks[2:]]
            # complexity filter is not part of fn_scan -> call the parser pieces directly
            L = orc.lib()
            s = seq.encode()
            import ctypes as C
            rle = C.create_string_buffer(len(s) + 2)
            pos = (C.c_uint64 * (len(s) + 2))()
            hl = L.orc_hpc_encode(s, C.c_size_t(len(s)), g["hpc"], rle, pos)
            om = (C.c_uint32 * max(hl, 1))(); op = (C.c_uint32 * max(hl, 1))(); od = (C.c_uint8 * max(hl, 1))()
            L.orc_minimizer_parse.restype = C.c_size_t
            n = L.orc_minimizer_parse(rle, C.c_size_t(hl), g["K"], C.c_float(g["density"]), None, C.c_size_t(0), om, op, od)
            got = [(om[i], op[i], od[i]) for i in range(n)]
            assert hl == int(toks[1]) and got == exp, key


def test_scan_without_end_trim(fn_golden):
    """MinimizerParser with _trimBps = 0 (GenerateGfa's LoadUnitigsFunctor) via refdrv fn_scan_notrim."""
    import ctypes as C
    L = orc.lib()
    L.orc_minimizer_parse_trim.restype = C.c_size_t
    n_end = 0
    for key, g in list(fn_golden["scan_notrim"].items()) + list(fn_golden["scan_notrim_block"].items()):
        for seq, out in zip(g["inputs"], g["outputs"]):
            toks = out.split()
            exp = [tuple(int(x) for x in t.split(":")) for t in toks[2:]]
            s = seq.encode()
            rle = C.create_string_buffer(len(s) + 2)
            pos = (C.c_uint64 * (len(s) + 2))()
            hl = L.orc_hpc_encode(s, C.c_size_t(len(s)), g["hpc"], rle, pos)
            om = (C.c_uint32 * max(hl, 1))(); op = (C.c_uint32 * max(hl, 1))(); od = (C.c_uint8 * max(hl, 1))()
            n = L.orc_minimizer_parse_trim(rle, C.c_size_t(hl), g["K"], C.c_float(g["density"]), None, C.c_size_t(0), C.c_size_t(0), om, op, od)
            got = [(om[i], op[i], od[i]) for i in range(n)]
            assert hl == int(toks[1]) and got == exp, key
            n_end += sum(1 for e in exp if e[1] == 0 or e[1] == hl - g["K"])
    assert n_end > 10      # the fixture does exercise the end positions


def test_apply_density_threshold(fn_golden):
    """Utils::applyDensityThreshold (Commons.hpp:2507-2550) via refdrv fn_density."""
    for dens, g in fn_golden["density"].items():
        for line, out in zip(g["inputs"], g["outputs"]):
            mins = np.array(line.split(), dtype=np.uint64).astype(np.uint32)
            exp = [int(x) for x in out.split()[1:]]
            assert orc.apply_density_threshold(mins, float(dens)).tolist() == exp, dens


def test_correction_scan(fn_golden):
    """ReadCorrection::ReadSelectionFunctor up to its sink, via refdrv fn_corrscan; the inclusive quality span must
    also be told apart from ReadSelection's on these reads (else the fixture pins nothing)."""
    differs = False
    for key, g in fn_golden["corrscan"].items():
        for seq, qual, out in zip(g["reads"], g["quals"], g["outputs"]):
            exp = [tuple(int(x) for x in t.split(":")) for t in out.split()[1:]]
            r = orc.correction_scan(seq.encode(), qual.encode(), K=g["K"], density=g["density"], hpc=bool(g["hpc"]))
            got = list(zip(r["minimizers"].tolist(), r["pos"].tolist(), r["dir"].tolist(), r["qual"].tolist()))
            assert got == exp, key
            assert r["mean_quality"] == 0.0 and not r["low_complexity"] and not r["low_quality"]
            if g["hpc"]:
                r0 = orc.read_selection(seq.encode(), qual.encode(), K=g["K"], density=g["density"], hpc=True)
                differs |= (not r0["low_complexity"]) and r0["qual"].tolist() != r["qual"].tolist()
    assert differs


@pytest.mark.parametrize("tag,K,dens,hpc", [("hpc_k15", 15, 0.005, True), ("nohpc_k15", 15, 0.005, False),
                                            ("hpc_k16", 16, 0.005, True), ("nohpc_k13", 13, 0.02, False)])
def test_edge_reads_fasta(tag, K, dens, hpc):
    seqs = H.read_fasta(os.path.join(H.GOLDEN, "edge", "edge.fasta"))
    rep = np.frombuffer(H.golden_bytes("edge", f"repetitiveMinimizers.{tag}.bin"), "<u4")
    got = b"".join(orc.read_selection(s, None, K=K, density=dens, hpc=hpc, repetitive=rep)["record"] for s in seqs)
    assert got == H.golden_bytes("edge", f"read_data_init.{tag}.txt")


@pytest.mark.parametrize("tag,hpc", [("fastq_hpc_k15", True), ("fastq_nohpc_k15", False)])
def test_edge_reads_fastq(tag, hpc):
    seqs, quals = H.read_fastq(os.path.join(H.GOLDEN, "edge", "edge.fastq"))
    rep = np.frombuffer(H.golden_bytes("edge", f"repetitiveMinimizers.{tag}.bin"), "<u4")
    got = b"".join(orc.read_selection(s, q, K=15, density=0.005, hpc=hpc, repetitive=rep)["record"]
                   for s, q in zip(seqs, quals))
    assert got == H.golden_bytes("edge", f"read_data_init.{tag}.txt")


def _check_set(name: str):
    m = H.load_manifest(name)
    seqs, quals = H.regenerate_reads(m)
    rep = np.frombuffer(H.golden_bytes(name, "repetitiveMinimizers.bin"), "<u4")
    recs = [orc.read_selection(s, quals[i] if quals else None, K=m["K"], density=m["density"], hpc=m["hpc"],
                               repetitive=rep) for i, s in enumerate(seqs)]
    assert b"".join(r["record"] for r in recs) == H.golden_bytes(name, "read_data_init.txt")
    # read_stats.txt (readSelection/ReadSelection.hpp:305-384)
    st = formats.parse_read_stats(H.golden_bytes(name, "read_stats.txt"))
    lens = np.array([len(s) for s in seqs], dtype=np.uint32)
    assert st["n_reads"] == len(seqs) and st["n_bases"] == int(lens.sum())
    assert st["n50"] == orc.lib().orc_compute_n50(lens.ctypes.data, len(lens))
    assert st["mean_length"] == orc.lib().orc_compute_mean_length(lens.ctypes.data, len(lens))
    assert st["n_minimizers"] == sum(len(r["minimizers"]) for r in recs)
    # purgePalindromes -> read_data_corrected.txt
    last_k = orc.lib().orc_compute_last_k(m["density"], st["n50"], 4, 0)
    purged = [orc.purge_palindrome(r["minimizers"], 4, last_k) for r in recs]
    offs = np.concatenate([[0], np.cumsum([len(p) for p in purged])]).astype(np.uint64)
    mins = np.concatenate(purged).astype(np.uint32)
    assert formats.write_minimizer_reads(mins, offs) == H.golden_bytes(name, "read_data_corrected.txt")
    # k-min-mer table, first pass
    t = orc.kminmer_count_first(mins, offs, m["k"], m["min_abundance"])
    exp_ab = np.fromfile(os.path.join(H.GOLDEN, name, "kminmerData_abundance.sorted.bin"), formats.ABUNDANCE_DTYPE)
    got_ab = formats.sorted_abundance_records(orc.table_abundance_records(t))
    assert np.array_equal(got_ab, exp_ab)
    exp_v = np.fromfile(os.path.join(H.GOLDEN, name, "kminmerData_min.sorted.bin"), "<u4").reshape(-1, m["k"])
    got_v = formats.sorted_vector_records(t["vecs"].astype("<u4").tobytes(), m["k"])
    assert np.array_equal(got_v, exp_v)
    return m, seqs, rep


def test_hifi_200_end_to_end():
    _check_set("hifi_200")


def test_ont_100_end_to_end():
    m, seqs, rep = _check_set("ont_100")
    # repetitive minimizer selection (readSelection/ReadSelection.hpp:497-561): the chosen minimizer must
    # have the maximal count at correction density (ties are order-unstable in the reference).
    import ctypes as C
    L = orc.lib()
    L.orc_minimizer_parse.restype = C.c_size_t
    counts: dict[int, int] = {}
    for s in seqs:
        n = len(s)
        om = (C.c_uint32 * n)(); op = (C.c_uint32 * n)(); od = (C.c_uint8 * n)()
        k = L.orc_minimizer_parse(s, C.c_size_t(n), 15, C.c_float(0.025), None, C.c_size_t(0), om, op, od)
        for i in range(k):
            counts[om[i]] = counts.get(om[i], 0) + 1
    n_keep = max(int(np.float32(0.00001) * len(counts)), 1)
    assert len(rep) == n_keep
    top = sorted(counts.values(), reverse=True)[n_keep - 1]
    assert all(counts[int(r)] >= top for r in rep)


def test_ont_rep_census():
    """determineRepetitiveMinimizers where the reference's pick is unambiguous (tests/golden/ont_rep: four minimizers,
    no count tied across the cut): the oracle's census must select exactly the reference's set."""
    import ctypes as C
    from metamdbg_amd import synth
    m = H.load_manifest("ont_rep")
    spec = H.spec_from_manifest(m)
    L = orc.lib()
    L.orc_minimizer_parse.restype = C.c_size_t
    genome = synth.genome_codes(spec)
    allm = []
    for r0 in range(0, spec.n_reads, 200):
        asc = synth.codes_to_ascii(synth.read_codes(spec, r0, min(r0 + 200, spec.n_reads), genome))
        for row in asc:
            sq = row.tobytes()
            n = len(sq)
            om = (C.c_uint32 * n)(); op = (C.c_uint32 * n)(); od = (C.c_uint8 * n)()
            k = L.orc_minimizer_parse(sq, C.c_size_t(n), 15, C.c_float(m["correction_density"]), None, C.c_size_t(0), om, op, od)
            allm.append(np.frombuffer(om, np.uint32, k).copy())
    vals, counts = np.unique(np.concatenate(allm), return_counts=True)
    n_keep = max(int(np.float32(0.00001) * np.float32(len(vals))), 1)
    order = np.sort(counts)[::-1]
    rep = np.frombuffer(H.golden_bytes("ont_rep", "repetitiveMinimizers.bin"), "<u4")
    assert len(vals) == m["n_distinct"] and n_keep == m["n_keep"] == len(rep)
    assert order[n_keep - 1] == m["cut_count"] > order[n_keep] == m["next_count"]
    assert set(rep.tolist()) == set(vals[counts >= order[n_keep - 1]].tolist())


@pytest.mark.parametrize("name", ["hifi_200", "ont_100"])
def test_reference_log_known_answers(name):
    """Numbers the reference itself logs for the fixture runs: solid / rescued counts, the abundance checksum
    sum(abundance * hash) (graph/CreateMdbg.cpp:3321) and the EdgeIndexer's edge count + checksum (:1184)."""
    m = H.load_manifest(name)
    log = m["reference_log"]
    ab = np.fromfile(os.path.join(H.GOLDEN, name, "kminmerData_abundance.sorted.bin"), formats.ABUNDANCE_DTYPE)
    assert int((ab["abundance"] > 1).sum()) == log["n_solid"] and int((ab["abundance"] == 1).sum()) == log["n_rescued"]
    with np.errstate(over="ignore"):
        ck = int((ab["lo"] * ab["abundance"].astype(np.uint64)).sum(dtype=np.uint64))
    assert ck == log["abundance_checksum"]
    vecs = np.fromfile(os.path.join(H.GOLDEN, name, "kminmerData_min.sorted.bin"), "<u4").reshape(-1, m["k"])
    hi, lo, eck = orc.edge_index(vecs)
    assert len(hi) == log["n_edges"] and eck == log["edge_checksum"]


@pytest.mark.parametrize("name", ["hifi_200", "ont_100"])
def test_unitig_edge_index_reference_count(name):
    """UnitigEdgeIndexer (CreateMdbg.hpp:4234-4512) on the reference's own unitigGraph.nodes.bin: the number of
    distinct unitig edges the reference logs ("Dereplicating unitig edges ... Done: N")."""
    m = H.load_manifest(name)
    mins, offs, idx = formats.parse_unitig_nodes(H.golden_bytes(name, "unitigGraph.nodes.bin"))
    assert len(idx) > 10 and (np.diff(offs.astype(np.int64)) >= m["k"]).all()
    hi, lo, ck = orc.unitig_edge_index(mins, offs, m["k"])
    assert len(hi) == m["reference_log"]["n_unitig_edges"]
    keys = np.stack([hi, lo], axis=1)
    assert len(np.unique(keys, axis=0)) == len(keys)


def test_device_hash_header_against_the_oracle(tmp_path):
    """csrc/murmur.hpp compiled for the host (tests/host/test_murmur_halves.cpp): the closed form of the minimizer hash equals
    the oracle's MurmurHash3_x64_128(8 bytes, seed 42), and the carry-less upper half the block kernel records candidates by is
    the upper half of the hash or one below it, so that no selected position can fail the candidate test."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    orc.lib()                                              # builds oracle/liboracle.so when missing
    exe = str(tmp_path / "murmur_halves")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", os.path.join(root, "tests", "host", "test_murmur_halves.cpp"), "-o", exe,
                    "-L", os.path.join(root, "oracle"), "-loracle", "-Wl,-rpath," + os.path.join(root, "oracle")], check=True)
    out = subprocess.run([exe, "8000000"], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("ok:"), out.stdout + out.stderr


def test_hifi_1m_manifest_is_self_consistent():
    """tests/golden/hifi_1m (BASELINE.json configs[1] at full size, digests only): the abundance checksum computed from the
    reference's records with the formula of graph/CreateMdbg.cpp:3321 (sum abundance * low word) is the number the reference
    itself logged for that table -- which pins mdbg_table_checksum's definition to the reference's log line -- and the
    counts add up."""
    import json
    path = os.path.join(H.GOLDEN, "hifi_1m", "manifest.json")
    if not os.path.exists(path):
        pytest.skip("tests/golden/hifi_1m/manifest.json not generated")
    g = json.load(open(path))
    log = g["reference_log"]
    assert g["abundance_checksum"] == log["abundance_checksum"]
    assert g["n_records"] == log["n_solid"] + log["n_rescued"]
    # read_data_init.txt: 13 bytes per read + 10 per minimizer; the purge only ever removes minimizers
    assert (g["read_data_init_bytes"] - 13 * g["n_reads"]) % 10 == 0
    assert (g["read_data_init_bytes"] - 13 * g["n_reads"]) // 10 >= g["n_corrected_minimizers"]
    assert g["n_reads"] == 1_000_000 and g["read_len"] == 10_000
