"""The oracle against the reference's own code run NOW (oracle/_ref/refdrv, built from /root/reference by oracle/Makefile)
on generated inputs -- beyond the committed vectors of tests/golden/fn_golden.json.  Skipped where the reference build is
absent.  MDBG_TEST_SEED=<n> draws different inputs (MDBG_TEST_SEED=time: from the clock); the seed is part of every
assertion message.  Round 1 ran about a hundred seeds of every test here without a difference."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import time

import numpy as np
import pytest

from oracle import pyoracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDRV = os.path.join(ROOT, "oracle", "_ref", "refdrv")
pytestmark = pytest.mark.skipif(not os.path.exists(REFDRV), reason="oracle/_ref/refdrv not built")
_seed = os.environ.get("MDBG_TEST_SEED", "20260927")
SEED = int(time.time()) % 1_000_000 if _seed == "time" else int(_seed)


def refdrv(args, lines):
    r = subprocess.run([REFDRV] + [str(a) for a in args], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-500:]
    return r.stdout.splitlines()


def rand_read(rng, n, kind):
    if kind == 0:
        codes = rng.integers(0, 4, n)
    elif kind == 1:                                   # homopolymer-rich
        codes = np.repeat(rng.integers(0, 4, n), rng.integers(1, 9, n))[:n]
    else:                                             # short tandem repeats
        unit = rng.integers(0, 4, int(rng.integers(1, 7)))
        codes = np.tile(unit, n // len(unit) + 1)[:n]
        flip = rng.random(n) < 0.02
        codes = np.where(flip, rng.integers(0, 4, n), codes)
    s = np.frombuffer(b"ACGT", dtype=np.uint8)[codes].copy()
    if kind == 0 and n > 10:
        for _ in range(int(rng.integers(0, 4))):      # N, n and lower case (Kmer.hpp:462: purely bitwise character handling)
            p = int(rng.integers(0, n))
            s[p] = rng.choice(np.frombuffer(b"NnacgtR", dtype=np.uint8))
    return bytes(s)


@pytest.mark.parametrize("K,density,hpc", [(15, 0.005, 1), (15, 0.05, 0), (13, 0.025, 0), (16, 0.2, 1), (7, 0.5, 1)])
def test_minimizer_parser_live(K, density, hpc):
    rng = np.random.default_rng(SEED + K)
    reads = [rand_read(rng, int(rng.integers(0, 4000)), int(rng.integers(0, 3))) for _ in range(60)]
    reads = [r for r in reads if r]                   # the driver reads one sequence per line
    L = orc.lib()
    L.orc_minimizer_parse.restype = C.c_size_t
    for trim, cmd in ((1, "fn_scan"), (0, "fn_scan_notrim")):
        out = refdrv([cmd, K, density, hpc], [r.decode() for r in reads])
        assert len(out) == len(reads)
        for seq, line in zip(reads, out):
            toks = line.split()
            exp = [tuple(int(x) for x in t.split(":")) for t in toks[2:]]
            rle = C.create_string_buffer(len(seq) + 2)
            pos = (C.c_uint64 * (len(seq) + 2))()
            hl = L.orc_hpc_encode(seq, C.c_size_t(len(seq)), hpc, rle, pos)
            cap = max(hl, 1)
            om = (C.c_uint32 * cap)(); op = (C.c_uint32 * cap)(); od = (C.c_uint8 * cap)()
            if trim:
                n = L.orc_minimizer_parse(rle, C.c_size_t(hl), K, C.c_float(density), None, C.c_size_t(0), om, op, od)
            else:
                L.orc_minimizer_parse_trim.restype = C.c_size_t
                n = L.orc_minimizer_parse_trim(rle, C.c_size_t(hl), K, C.c_float(density), None, C.c_size_t(0), C.c_size_t(0), om, op, od)
            got = [(om[i], op[i], od[i]) for i in range(n)]
            assert hl == int(toks[1]) and got == exp, (SEED, cmd, K, density, hpc, seq[:60])


@pytest.mark.parametrize("k", [3, 4, 5, 8, 11, 21])
def test_kminmer_normalize_hash_live(k):
    rng = np.random.default_rng(SEED + 100 + k)
    vecs = []
    for _ in range(200):
        v = rng.integers(0, 2 ** 32, k, dtype=np.uint64).astype(np.uint32)
        t = int(rng.integers(0, 4))
        if t == 0:
            v = np.concatenate([v[: k // 2], v[: k - k // 2][::-1]])[:k]     # palindromes and near-palindromes
        elif t == 1:
            v[:] = v[0]
        vecs.append(v)
    out = refdrv(["fn_kminmer", k], [" ".join(str(int(x)) for x in v) for v in vecs])
    for v, line in zip(vecs, out):
        o = [int(x) for x in line.split()]
        rev, cvec, hi, lo = orc.kminmer_normalize_hash(v)
        assert [rev, hi, lo] + cvec.tolist() == o, (SEED, k, v.tolist())


@pytest.mark.parametrize("first_k,last_k", [(4, 6), (4, 11), (3, 9), (5, 21)])
def test_purge_palindrome_live(first_k, last_k):
    rng = np.random.default_rng(SEED + 200 + last_k)
    lines = []
    for _ in range(150):
        n = int(rng.integers(0, 60))
        alphabet = int(rng.integers(2, 8))            # few distinct minimizers: palindromic windows everywhere
        v = rng.integers(0, alphabet, n)
        if n > 8 and rng.integers(0, 2):
            h = int(rng.integers(2, n // 2))
            a = int(rng.integers(0, n - 2 * h + 1))
            v[a + h: a + 2 * h] = v[a: a + h][::-1]
        lines.append(" ".join(str(int(x)) for x in v))
    lines = [ln for ln in lines if ln]
    out = refdrv(["fn_purge", first_k, last_k], lines)
    for line, o in zip(lines, out):
        got = orc.purge_palindrome([int(x) for x in line.split()], first_k, last_k).tolist()
        assert got == [int(x) for x in o.split()], (SEED, first_k, last_k, line)


def test_murmur_and_density_live():
    rng = np.random.default_rng(SEED + 300)
    vals = [int(x) for x in rng.integers(0, 2 ** 63, 300, dtype=np.uint64)] + [0, 1, 2 ** 30 - 1, 2 ** 32 - 1, 2 ** 64 - 1]
    out = refdrv(["fn_murmur"], [str(v) for v in vals])
    assert [int(x) for x in out] == [orc.kmer_hash(v) for v in vals], SEED
    for d in (0.005, 0.025, 0.3):
        lines = [" ".join(str(int(x)) for x in rng.integers(0, 2 ** 32, int(rng.integers(1, 400)), dtype=np.uint64)) for _ in range(20)]
        out = refdrv(["fn_density", d], lines)
        for line, o in zip(lines, out):
            m = np.array([int(x) for x in line.split()], dtype=np.uint32)
            keep = orc.apply_density_threshold(m, d).tolist()          # indices kept; the driver prints their number first
            assert keep == [int(x) for x in o.split()[1:]], (SEED, d)


@pytest.mark.parametrize("kind", ["hifi_fasta", "hifi_fastq", "ont_fastq"])
def test_pipeline_live(kind, tmp_path):
    """readSelection + graph --firstpass of the reference on a read set drawn now (repeats, low-complexity reads, short
    reads, skewed base composition) against the oracle: read_data_init.txt byte for byte, read_data_corrected.txt, the table."""
    from metamdbg_amd import formats
    rng = np.random.default_rng(SEED + 400 + len(kind))
    hpc = kind != "ont_fastq"
    with_q = kind != "hifi_fasta"
    p = np.array([0.3, 0.2, 0.2, 0.3]) if kind == "hifi_fastq" else np.full(4, 0.25)
    genome = rng.choice(4, size=40_000, p=p)
    genome[5000:5400] = np.tile([0, 1], 200)                              # a low-complexity stretch
    genome[9000:9300] = 2                                                  # a long homopolymer
    seqs, quals = [], []
    for i in range(150):
        L = int(rng.integers(20, 9000)) if i % 10 else int(rng.integers(0, 70))
        a = int(rng.integers(0, len(genome) - L))
        c = genome[a:a + L].copy()
        err = rng.random(L) < (0.002 if hpc else 0.03)
        c = np.where(err, rng.integers(0, 4, L), c)
        if rng.integers(0, 2):
            c = (c[::-1] ^ 2)                                              # reverse complement in the A0 C1 T2 G3 code
        seqs.append(bytes(np.frombuffer(b"ACTG", dtype=np.uint8)[c]))
        quals.append(bytes((rng.integers(1, 60, L) + 33).astype(np.uint8)))
    path = str(tmp_path / ("r.fastq" if with_q else "r.fasta"))
    with open(path, "wb") as f:
        for i, s in enumerate(seqs):
            if not s:
                s, quals[i] = b"A", b"I"                                   # kseq skips nothing, but an empty record has no line to parse
                seqs[i] = s
            f.write((b"@r%d\n" % i + s + b"\n+\n" + quals[i] + b"\n") if with_q else (b">r%d\n" % i + s + b"\n"))
    P = formats.Parameters(minimizer_size=15, kminmer_size=4, density=0.005, first_k=4, prev_k=4, hpc=hpc, data_type=0 if hpc else 1,
                           correction_density=0.025)
    tmp = str(tmp_path / "out" / "tmp")
    for d in ("", "filter", "smallContigs", "checkpoints"):
        os.makedirs(os.path.join(tmp, d), exist_ok=True)
    P.save(os.path.join(tmp, "parameters.gz"))
    open(os.path.join(tmp, "input.txt"), "w").write(path + "\n")
    rs = [REFDRV, "readSelection", tmp, tmp + "/read_data_init.txt", tmp + "/input.txt", "--threads", "1", "--min-read-quality", "0.000000"]
    subprocess.run(rs + ([] if hpc else ["--skip-correction"]), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
    subprocess.run([REFDRV, "graph", tmp, "--threads", "1", "--min-abundance", "0", "--firstpass"], check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL, timeout=300)
    rd = lambda n: open(os.path.join(tmp, n), "rb").read()
    rep = np.frombuffer(rd("repetitiveMinimizers.bin"), "<u4")
    recs = [orc.read_selection(s, quals[i] if with_q else None, K=15, density=0.005, hpc=hpc, repetitive=rep) for i, s in enumerate(seqs)]
    assert b"".join(r["record"] for r in recs) == rd("read_data_init.txt"), (SEED, kind)
    st = formats.parse_read_stats(rd("read_stats.txt"))
    last_k = orc.lib().orc_compute_last_k(0.005, st["n50"], 4, 0)
    purged = [orc.purge_palindrome(r["minimizers"], 4, last_k) for r in recs]
    offs = np.concatenate([[0], np.cumsum([len(x) for x in purged])]).astype(np.uint64)
    mins = np.concatenate(purged).astype(np.uint32) if purged else np.zeros(0, np.uint32)
    assert formats.write_minimizer_reads(mins, offs) == rd("read_data_corrected.txt"), (SEED, kind)
    t = orc.kminmer_count_first(mins, offs, 4, 0)
    assert np.array_equal(formats.sorted_abundance_records(orc.table_abundance_records(t)), formats.sorted_abundance_records(rd("kminmerData_abundance.txt"))), (SEED, kind)
    assert np.array_equal(formats.sorted_vector_records(t["vecs"].astype("<u4").tobytes(), 4), formats.sorted_vector_records(rd("kminmerData_min.txt"), 4)), (SEED, kind)


def test_multik_loop_live(tmp_path):
    """The reference's multi-k loop (graph -> contig -> toMinspace per k, as tests/golden/make_golden.py drives it) on reads
    drawn now; every k > firstK: previous table + unitig overlay, refined count / index and the small-contig branch of the
    oracle against what the reference wrote."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden as mg
    from metamdbg_amd import formats, synth
    from tests import multik_fixture as mk
    spec = synth.hifi_spec(160, seed=SEED % 100000 + 7, coverage=25.0)
    fasta = str(tmp_path / "reads.fasta")
    synth.write_fasta(fasta, spec)
    params = formats.Parameters(minimizer_size=15, kminmer_size=4, density=0.005, first_k=4, prev_k=4, last_k=0, hpc=True, data_type=0)
    tmp = mg.run_ref_pipeline(str(tmp_path / "ref"), fasta, params, graph=False)
    root = str(tmp_path / "fx")
    mg.run_ref_multik(tmp, params, 10, os.path.join(root, "live"), dict(kind="hifi", seed=spec.seed))
    for k in range(5, 11):
        fx = mk.load("live", k, root=root)
        P = fx["params"]
        prev = orc.PrevAbundance(fx["prev_records"])
        prev.overlay_unitigs(fx["prev_unitigs"], P.prev_k)
        (rm, ro), (um, uo) = fx["reads"], fx["unitigs"]
        allm = np.concatenate([rm, um]); alloff = np.concatenate([ro, ro[-1] + uo[1:]])
        t = (orc.kminmer_count_refined if k == P.first_k + 1 else orc.kminmer_index)(allm, alloff, k, prev)
        got = formats.sorted_abundance_records(orc.table_abundance_records(t))
        assert np.array_equal(got, fx["abundance_sorted"]), (SEED, k)
        if fx["min_sorted"] is not None:
            assert np.array_equal(formats.sorted_vector_records(t["vecs"].astype("<u4").tobytes(), k), fx["min_sorted"]), (SEED, k)
        flags = orc.small_contigs(um, uo, k, P.prev_k, prev) if k > 8 else np.zeros(len(uo) - 1, np.uint8)
        mine = sorted((0, tuple(int(x) for x in um[int(uo[i]): int(uo[i + 1])])) for i in np.nonzero(flags)[0])
        assert mine == mk.small_contig_records(fx["small_contigs"]), (SEED, k)


def test_graph_stage_on_foreign_tables_live(tmp_path):
    """The harness of the hand-over test (tests/handover.py) checked on the CPU: the reference's multi-k loop run twice,
    once as it is and once with every `graph` split into (tables written by someone else: here the reference's own records
    in shuffled order) + `refdrv graph_from_tables` (the reference's graph stage alone; at k >= firstK+2 its in-memory
    table filled from the 20-byte records).  Everything the next stage reads must come out the same.  The GPU test
    (tests/test_gpu_tool.py::test_handover_into_reference_graph_stage) swaps the stand-in producer for mdbg_tool."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden as mg
    from metamdbg_amd import formats, synth
    from tests import handover as ho
    spec = synth.hifi_spec(160, seed=SEED % 100000 + 13, coverage=25.0)
    fasta = str(tmp_path / "reads.fasta")
    synth.write_fasta(fasta, spec)
    params = formats.Parameters(minimizer_size=15, kminmer_size=4, density=0.005, first_k=4, prev_k=4, last_k=0, hpc=True, data_type=0)
    t_ref = mg.run_ref_pipeline(str(tmp_path / "ref"), fasta, params, graph=False)
    t_hyb = mg.run_ref_pipeline(str(tmp_path / "hyb"), fasta, params, graph=False)
    last_k = 9
    ho.run_loop(t_ref, params, last_k, ho.reference_graph, str(tmp_path / "snap_ref"))
    producer = ho.shuffled_reference_tables(str(tmp_path / "scratch"), seed=SEED)
    ho.run_loop(t_hyb, params, last_k, ho.tables_then_reference_graph_stage(producer), str(tmp_path / "snap_hyb"))
    seen = ho.compare_dirs(str(tmp_path / "snap_ref"), str(tmp_path / "snap_hyb"), 4, last_k)
    assert seen["graph_files"] >= 4 * (last_k - 3) - 4 and seen["next_inputs"] == 3 * (last_k - 4), seen
