"""The C ABI as a drop-in INSIDE the reference: oracle/_ref/refdrv_hip is the reference's own code compiled where it lies
plus oracle/ref_binding.cpp -- classes DERIVED from the reference's ReadSelection and CreateMdbg in which only what INTEGRATION.md sections
1 and 2 name is replaced by calls through include/mdbg_hip.h.  Argument parsing, ReadParserParallel + kseq, the ordered record writer
(writeRead), computeReadStats, computeLastK, Tool::end, and the whole graph stage behind the tables are the reference's, unchanged.

Each test runs the modified and the unmodified reference tool side by side on the same input and compares their files."""
from __future__ import annotations

import os
import struct
import sys

import numpy as np
import pytest

from metamdbg_amd import formats, synth
from tests import helpers as H
from tests.test_gpu_tool import REFDRV, fbytes, make_tmp, run

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(REFDRV), "refdrv_hip")), reason="oracle/_ref/refdrv_hip not built")]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDRV_HIP = os.path.join(ROOT, "oracle", "_ref", "refdrv_hip")


def _corrected_records(raw: bytes) -> list[bytes]:
    out, o = [], 0
    while o < len(raw):
        k = struct.unpack_from("<I", raw, o)[0]
        out.append(raw[o:o + 5 + 4 * k])
        o += 5 + 4 * k
    return sorted(out)


def _read_selection(exe, cmd, tmp, threads, extra=(), env=None):
    run(exe, cmd, tmp, os.path.join(tmp, "read_data_init.txt"), os.path.join(tmp, "input.txt"), "--threads", str(threads), *extra, env=env)


def _stats_equal(a: bytes, b: bytes, exact: bool) -> None:
    """read_stats.txt: byte for byte when the binding's parser ran on one thread (reads arrive in read order, the long-double quality sum
    is accumulated in the reference's order); with several threads the reads arrive out of order and the mean quality -- a float of that
    sum -- may differ in its last bit, as it does between two runs of the reference itself."""
    if exact:
        assert a == b
        return
    x, y = formats.parse_read_stats(a), formats.parse_read_stats(b)
    for key in ("n_reads", "n50", "density", "n_bases", "mean_length", "n_minimizers"):
        assert x[key] == y[key], key
    assert abs(x["avg_quality"] - y["avg_quality"]) <= 1e-5 * abs(y["avg_quality"])


@pytest.mark.parametrize("batch,threads", [("200000", 4), ("67108864", 1)])
def test_reference_read_selection_with_the_binding_hifi_fastq(tmp_path, batch, threads):
    """HiFi preset, FASTQ with qualities, a quality threshold that removes reads, soft-masked stretches: read_data_init.txt and
    read_stats.txt byte for byte, read_data_corrected.txt as a multiset (the reference writes it in thread-completion order).
    (No N: the unmodified reference corrupts its heap on a read that holds one -- kmerCounts[-1], ReadSelection.hpp:1196)"""
    rng = np.random.default_rng(71)
    genome = np.repeat(synth.CODE2ASCII[rng.integers(0, 4, 50000)], rng.choice([1, 1, 2, 5], 50000))
    fq = str(tmp_path / "r.fastq")
    with open(fq, "wb") as f:
        for i in range(400):
            a = int(rng.integers(0, len(genome) - 12000)); L = int(rng.integers(100, 12000))
            s = genome[a:a + L].copy()
            if i % 23 == 0:
                b = int(rng.integers(0, L)); s[b:b + 40] |= 0x20
            q = (rng.integers(2, 60 if i % 5 else 12, L) + 33).astype(np.uint8)
            f.write(b"@r%d\n" % i + bytes(s) + b"\n+\n" + bytes(q) + b"\n")
    P = formats.Parameters(minimizer_size=15, kminmer_size=4, density=0.005, first_k=4, prev_k=4, hpc=True, data_type=0)
    t_ref, t_hip = make_tmp(tmp_path / "ref", P, [fq]), make_tmp(tmp_path / "hip", P, [fq])
    q = ["--min-read-quality", "9.5"]
    _read_selection(REFDRV, "readSelection", t_ref, 1, q)
    _read_selection(REFDRV_HIP, "readSelection_hip", t_hip, threads, q, env={"MDBG_BINDING_BATCH_BASES": batch})
    for name in ("read_data_init.txt", "repetitiveMinimizers.bin"):
        assert fbytes(t_hip, name) == fbytes(t_ref, name), name
    _stats_equal(fbytes(t_hip, "read_stats.txt"), fbytes(t_ref, "read_stats.txt"), exact=threads == 1)
    assert _corrected_records(fbytes(t_hip, "read_data_corrected.txt")) == _corrected_records(fbytes(t_ref, "read_data_corrected.txt"))
    st = formats.parse_read_stats(fbytes(t_ref, "read_stats.txt"))
    assert st["n_reads"] == 400 and 0 < st["n_minimizers"]
    assert os.path.getsize(os.path.join(t_hip, "perf.bin")) == 16


def test_reference_read_selection_with_the_binding_ont_census(tmp_path):
    """ONT preset (no HPC, repetitive-minimizer census, --skip-correction) on the read set whose pick is unambiguous
    (tests/golden/ont_rep): the census through the library must choose what the reference's own counting chooses, and the file
    filtered by it must be the reference's."""
    m = H.load_manifest("ont_rep")
    fastq = str(tmp_path / "ont_rep.fastq")
    synth.write_fasta(fastq, H.spec_from_manifest(m))
    P = formats.Parameters(minimizer_size=15, kminmer_size=4, density=0.005, first_k=4, prev_k=4, hpc=False, data_type=1, correction_density=0.025)
    t_ref, t_hip = make_tmp(tmp_path / "ref", P, [fastq]), make_tmp(tmp_path / "hip", P, [fastq])
    extra = ["--min-read-quality", "0.0", "--skip-correction"]
    _read_selection(REFDRV, "readSelection", t_ref, 1, extra)
    _read_selection(REFDRV_HIP, "readSelection_hip", t_hip, 4, extra, env={"MDBG_BINDING_BATCH_BASES": "3000000"})
    rep = lambda t: set(np.frombuffer(fbytes(t, "repetitiveMinimizers.bin"), "<u4").tolist())
    assert rep(t_hip) == rep(t_ref) and len(rep(t_ref)) == m["n_keep"]
    assert fbytes(t_hip, "read_data_init.txt") == fbytes(t_ref, "read_data_init.txt")
    _stats_equal(fbytes(t_hip, "read_stats.txt"), fbytes(t_ref, "read_stats.txt"), exact=False)
    assert _corrected_records(fbytes(t_hip, "read_data_corrected.txt")) == _corrected_records(fbytes(t_ref, "read_data_corrected.txt"))


@pytest.mark.parametrize("last_k", [9, 100])
def test_reference_multi_k_loop_with_the_binding(tmp_path, last_k):
    """The reference's multi-k loop (graph -> contig -> toMinspace, k = 4 .. 9) twice: with its own `graph`, and with `graph_hip` -- the
    reference's CreateMdbg whose tables come from the library, its graph stage on top of them in the same command.  Unitig graph
    files and the inputs of every next k byte-equal, tables equal as multisets, at every k.  last_k = 100: the loop of the reference's
    default `asm` for 10 kb reads (no --max-k: lastK = N50 x density x 2, Commons.hpp:1726-1741), 97 passes."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden as mg
    from tests import handover as ho
    spec = synth.hifi_spec(260, seed=93, coverage=30.0)
    params = formats.Parameters(minimizer_size=15, kminmer_size=4, density=0.005, first_k=4, prev_k=4, last_k=0, hpc=True, data_type=0)
    reads = str(tmp_path / "reads.fasta")
    synth.write_fasta(reads, spec)
    t_ref = mg.run_ref_pipeline(str(tmp_path / "ref"), reads, params, graph=False)
    t_hip = mg.run_ref_pipeline(str(tmp_path / "hip"), reads, params, graph=False)
    ho.run_loop(t_ref, params, last_k, ho.reference_graph, str(tmp_path / "snap_ref"))
    ho.run_loop(t_hip, params, last_k, lambda tmp, k, first_k: run(REFDRV_HIP, "graph_hip", *ho.graph_args(tmp, k, first_k)), str(tmp_path / "snap_hip"))
    seen = ho.compare_dirs(str(tmp_path / "snap_ref"), str(tmp_path / "snap_hip"), 4, last_k)
    assert seen["graph_files"] >= 4 * (last_k - 4) and seen["next_inputs"] == 3 * (last_k - 4) and seen["tables"] == last_k - 3, seen


def test_correction_scan_inside_the_reference(tmp_path):
    """SURVEY 8(f) N1 inside the reference's process: `refdrv_hip fn_corrscan_hip` -- the compute of ReadCorrection::ReadSelectionFunctor
    (ReadCorrection.hpp:2298-2342: EncoderRLE, MinimizerParser::parse at the correction density, getMinQuality over [rle[pos], rle[pos+l-1]])
    as ONE mdbg_scan with quality_window = 1 -- prints what `refdrv fn_corrscan`, the reference's own functor code, prints: on the
    homopolymer-rich golden reads at three (l, density) settings with and without HPC, and on the 100 ONT reads of tests/golden/ont_100
    at the correction density 0.025.  (The functor's sink -- TurboPFor + BGZF partitions -- does not link here: oracle/ref_binding.cpp.)"""
    import json
    import subprocess
    fn = json.load(open(os.path.join(H.GOLDEN, "fn", "fn_golden.json")))["corrscan"]
    cases = [(c["K"], c["density"], c["hpc"], [f"{r} {q}" for r, q in zip(c["reads"], c["quals"])]) for c in fn.values()]
    m = H.load_manifest("ont_100")
    spec = H.spec_from_manifest(m)
    asc = synth.codes_to_ascii(synth.read_codes(spec, 0, spec.n_reads))
    qual = synth.read_qualities(spec, 0, spec.n_reads)
    cases.append((15, 0.025, 0, [asc[i].tobytes().decode() + " " + qual[i].tobytes().decode() for i in range(spec.n_reads)]))
    n_min = 0
    for K, dens, hpc, lines in cases:
        text = "\n".join(lines) + "\n"
        a = subprocess.run([REFDRV, "fn_corrscan", str(K), str(dens), str(hpc)], input=text, capture_output=True, text=True, timeout=300)
        b = subprocess.run([REFDRV_HIP, "fn_corrscan_hip", str(K), str(dens), str(hpc)], input=text, capture_output=True, text=True, timeout=300)
        assert a.returncode == 0 and b.returncode == 0, (a.stderr[-500:], b.stderr[-500:])
        assert a.stdout == b.stdout, (K, dens, hpc)
        n_min += sum(int(l.split(" ", 1)[0]) for l in a.stdout.splitlines())
    assert n_min > 40_000


def test_edge_index_inside_the_reference(tmp_path):
    """SURVEY 8(f) N2 inside the reference's process: after `graph_hip --firstpass` left the library's tables in the directory,
    `refdrv_hip edges_hip` runs the REFERENCE's own EdgeIndexer (graph/CreateMdbg.hpp:4010-4230, constructed on its CreateMdbg as
    CreateMdbg::indexEdges does, graph/CreateMdbg.cpp:1181-1183) on kminmerData_min.txt and mdbg_edge_index on the same vectors:
    every 128-bit identity, the count and the checksum the reference logs must agree -- and be what the reference logged when it made the
    fixture (tests/golden/hifi_200, ont_100)."""
    for name, extra in (("hifi_200", []), ("ont_100", ["--skip-correction"])):
        m = H.load_manifest(name)
        reads = str(tmp_path / f"{name}.fx")
        synth.write_fasta(reads, H.spec_from_manifest(m))
        P = formats.Parameters(minimizer_size=15, kminmer_size=4, density=0.005, first_k=4, prev_k=4, hpc=bool(m["hpc"]), data_type=0 if m["hpc"] else 1,
                               correction_density=0.025)
        tmp = make_tmp(tmp_path / name, P, [reads])
        _read_selection(REFDRV, "readSelection", tmp, 1, ["--min-read-quality", "0.0"] + extra)
        run(REFDRV_HIP, "graph_hip", tmp, "--threads", "2", "--min-abundance", "0", "--firstpass")
        r = run(REFDRV_HIP, "edges_hip", tmp, "--threads", "2", "--min-abundance", "0", "--firstpass")
        log = m["reference_log"]
        assert f"reference EdgeIndexer {log['n_edges']} keys, checksum {log['edge_checksum']}; mdbg_edge_index {log['n_edges']} keys, checksum {log['edge_checksum']}; equal" in r.stdout, r.stdout
        # ... and UnitigEdgeIndexer over the unitigs the graph stage left, against mdbg_unitig_edge_index
        assert f"reference UnitigEdgeIndexer {log['n_unitig_edges']} keys" in r.stdout and f"mdbg_unitig_edge_index {log['n_unitig_edges']} keys; equal" in r.stdout, r.stdout
