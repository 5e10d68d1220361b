#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE's own code.

Runs in the build container only (needs /root/reference, compiled in place into
oracle/_ref/refdrv by oracle/Makefile).  The fixtures are data: inputs (seeds / literal
sequences) and the reference's outputs.  No reference source is stored here.

    python tests/golden/make_golden.py            # regenerates everything

Fixture sets:
  hifi_200/   200 x 10 kb synthetic HiFi reads (FASTA), HPC on, l=15, density 0.005, k=4
  ont_100/    100 x 20 kb synthetic ONT-like reads (FASTQ), no HPC, --skip-correction
  edge/       hand-made reads: N, lowercase, < K, low complexity, homopolymers, K=16
  fn/         function-level known answers: purgePalindrome, normalize+hash128, murmur, lastK
"""
from __future__ import annotations

import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from metamdbg_amd import formats, synth  # noqa: E402

REFDRV = os.path.join(ROOT, "oracle", "_ref", "refdrv")


def run_ref_pipeline(workdir: str, fasta: str, params: formats.Parameters, threads: int = 1,
                     extra_rs: list[str] | None = None, graph: bool = True) -> str:
    """readSelection (+ graph --firstpass) exactly as AssemblyPipeline invokes them
    (pipeline/AssemblyPipeline.hpp:733-737, :770-783).  Returns the tmp dir."""
    tmp = os.path.join(workdir, "tmp")
    for d in ("", "filter", "smallContigs", "checkpoints"):
        os.makedirs(os.path.join(tmp, d), exist_ok=True)
    params.save(os.path.join(tmp, "parameters.gz"))
    with open(os.path.join(tmp, "input.txt"), "w") as f:
        f.write(fasta + "\n")
    cmd = [REFDRV, "readSelection", tmp, os.path.join(tmp, "read_data_init.txt"), os.path.join(tmp, "input.txt"),
           "--threads", str(threads), "--min-read-quality", "0.000000"] + (extra_rs or [])
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    if graph:
        cmd = [REFDRV, "graph", tmp, "--threads", str(threads), "--min-abundance", "0", "--firstpass"]
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return tmp


def sha256(path: str) -> str:
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def log_known_answers(tmp: str, start: int = 0) -> dict:
    """Numbers the reference logs while running `graph` (metaMDBG.log next to tmp/, from byte `start` on): the EdgeIndexer's
    edge count and checksum (graph/CreateMdbg.cpp:1184) and the abundance checksum sum(abundance * hash) (:3321, :3397)."""
    import re
    log = open(os.path.join(os.path.dirname(tmp), "metaMDBG.log")).read()[start:]
    out = {}
    m = re.search(r"Dereplicating edges.*?\n\s*Done: (\d+) (\d+) ", log, flags=re.S)
    if m:
        out["n_edges"], out["edge_checksum"] = int(m.group(1)), int(m.group(2))
    m = re.search(r"Dereplicating unitig edges.*?\n\s*Done: (\d+) ", log, flags=re.S)   # UnitigEdgeIndexer (CreateMdbg.hpp:4234-4512)
    if m:
        out["n_unitig_edges"] = int(m.group(1))
    m = re.search(r"Checksum kminmer abundance: (\d+)", log)
    if m:
        out["abundance_checksum"] = int(m.group(1))
    for key, pat in (("n_solid", r"Nb solid kminmers: (\d+)"), ("n_rescued", r"Nb rescued kminmers: (\d+)")):
        m = re.search(pat, log)
        if m:
            out[key] = int(m.group(1))
    return out


def store_outputs(tmp: str, dst: str, k: int, manifest: dict) -> None:
    os.makedirs(dst, exist_ok=True)
    if os.path.exists(os.path.join(tmp, "kminmerData_abundance.txt")):
        manifest = dict(manifest, reference_log=log_known_answers(tmp))
    for name in ("read_data_init.txt", "read_stats.txt", "repetitiveMinimizers.bin", "parameters.gz"):
        shutil.copy(os.path.join(tmp, name), os.path.join(dst, name))
    p = os.path.join(tmp, "read_data_corrected.txt")
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, "read_data_corrected.txt"))  # threads=1 -> read order
    p = os.path.join(tmp, "unitigGraph.nodes.bin")          # the reference's unitigs (data): input of the unitig-edge index
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, "unitigGraph.nodes.bin"))
    ab = os.path.join(tmp, "kminmerData_abundance.txt")
    if os.path.exists(ab):
        raw = open(ab, "rb").read()
        formats.sorted_abundance_records(raw).tofile(os.path.join(dst, "kminmerData_abundance.sorted.bin"))
        raw = open(os.path.join(tmp, "kminmerData_min.txt"), "rb").read()
        formats.sorted_vector_records(raw, k).astype("<u4").tofile(os.path.join(dst, "kminmerData_min.sorted.bin"))
    with open(os.path.join(dst, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


def make_hifi(work: str) -> None:
    spec = synth.hifi_spec(200, seed=7, coverage=40.0)
    fasta = os.path.join(work, "hifi_200.fasta")
    synth.write_fasta(fasta, spec)
    params = formats.Parameters(minimizer_size=15, kminmer_size=4, density=0.005, first_k=4, prev_k=4,
                                last_k=0, hpc=True, data_type=0)
    tmp = run_ref_pipeline(os.path.join(work, "hifi"), fasta, params)
    store_outputs(tmp, os.path.join(HERE, "hifi_200"), 4, dict(
        kind="hifi", n_reads=spec.n_reads, read_len=spec.read_len, seed=spec.seed, sub_rate=spec.sub_rate,
        species_len=spec.species_len, species_weight=spec.species_weight, fasta_sha256=sha256(fasta),
        K=15, density=0.005, hpc=True, k=4, min_abundance=0))


def make_hifi_1m(work: str, threads: int = 8) -> None:
    """BASELINE.json configs[1] at its stated size: 1 M synthetic HiFi reads x 10 kb (50x over the metagenome of
    synth.hifi_spec), readSelection + graph --firstpass by the reference's own code.  10 Gbp of FASTA and 400 MB of products
    do not go into the repository: only their digests do (sha256 of read_data_init.txt -- it is written in read order whatever
    the thread count --, order-independent digests of the files the reference writes in thread order: formats.
    minimizer_reads_digest, formats.table_digests) together with the counts and the checksum the reference logs.  The reads
    are regenerated from the seed (on the device, bit-identical: tests/test_gpu_parity.py::test_hifi_1m_digests)."""
    spec = synth.hifi_spec(1_000_000, seed=42, read_len=10_000, coverage=50.0)
    fasta = os.path.join(work, "hifi_1m.fasta")
    synth.write_fasta(fasta, spec)
    params = formats.Parameters(minimizer_size=15, kminmer_size=4, density=0.005, first_k=4, prev_k=4,
                                last_k=0, hpc=True, data_type=0)
    tmp = run_ref_pipeline(os.path.join(work, "hifi_1m"), fasta, params, threads=threads)
    # a second graph --firstpass run in the directory of the next k loads the table again and logs its checksum; here the
    # checksum is computed from the records with the formula of graph/CreateMdbg.cpp:3321 (abundance * hash, low 64 bits)
    rec = formats.parse_abundance_table(open(os.path.join(tmp, "kminmerData_abundance.txt"), "rb").read())
    with np.errstate(over="ignore"):
        checksum = int((rec["abundance"].astype(np.uint64) * rec["lo"]).sum(dtype=np.uint64))
    cm, co = formats.parse_minimizer_reads(open(os.path.join(tmp, "read_data_corrected.txt"), "rb").read())
    manifest = dict(
        kind="hifi", config="BASELINE.json configs[1]", n_reads=spec.n_reads, read_len=spec.read_len, seed=spec.seed,
        sub_rate=spec.sub_rate, species_len=spec.species_len, species_weight=spec.species_weight, fasta_sha256=sha256(fasta),
        K=15, density=0.005, hpc=True, k=4, min_abundance=0, reference_threads=threads,
        read_data_init_sha256=sha256(os.path.join(tmp, "read_data_init.txt")),
        read_data_init_bytes=os.path.getsize(os.path.join(tmp, "read_data_init.txt")),
        read_stats_hex=open(os.path.join(tmp, "read_stats.txt"), "rb").read().hex(),
        read_data_corrected_digest=formats.minimizer_reads_digest(cm, co), n_corrected_minimizers=int(len(cm)),
        n_records=int(len(rec)), abundance_checksum=checksum, sum_abundance=int(rec["abundance"].astype(np.uint64).sum()),
        reference_log=log_known_answers(tmp),
        **formats.table_digests(rec, open(os.path.join(tmp, "kminmerData_min.txt"), "rb").read(), 4))
    dst = os.path.join(HERE, "hifi_1m")
    os.makedirs(dst, exist_ok=True)
    with open(os.path.join(dst, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


def _graph_until_tables(tmp: str, threads: int, first_pass: bool, timeout: int = 3600) -> None:
    """The reference's `graph` up to the moment its tables are written and closed: the log line "Nb small contigs written" follows
    `_kminmerAbundanceFile.close()` (graph/CreateMdbg.cpp:466-470); what comes after is graph construction (minutes at a million
    reads, out of this repository's scope) and the process -- this very one, by its handle -- is ended there."""
    import time
    log_path = os.path.join(os.path.dirname(tmp), "metaMDBG.log")
    start = os.path.getsize(log_path) if os.path.exists(log_path) else 0
    cmd = [REFDRV, "graph", tmp, "--threads", str(threads)] + (["--min-abundance", "0", "--firstpass"] if first_pass else [])
    proc = subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    t0 = time.time()
    try:
        while proc.poll() is None:
            if os.path.exists(log_path):
                with open(log_path, "rb") as f:
                    f.seek(start)
                    if b"Nb small contigs written" in f.read():
                        break
            if time.time() - t0 > timeout:
                raise subprocess.TimeoutExpired(cmd, timeout)
            time.sleep(0.05)
        else:
            # the process ended by itself: fine if its tables were closed first (with empty unitig files the graph construction that
            # follows has nothing to stand on and may simply crash -- it is not what is being recorded here)
            done = False
            if os.path.exists(log_path):
                with open(log_path, "rb") as f:
                    f.seek(start)
                    done = b"Nb small contigs written" in f.read()
            if not done:
                raise subprocess.CalledProcessError(proc.returncode, cmd)
    finally:
        if proc.poll() is None:
            proc.kill()
            proc.wait()


def make_hifi_1m_multik(work: str, threads: int = 8, last_k: int = 11) -> None:
    """BASELINE.json configs[2]'s loop k = 4 .. 11 at configs[1]'s size by the reference's own code, in the benchmark mode SURVEY.md 8(d)
    defines (reads only: no unitig_data.txt, the previous table is the pass's own k - 1 output): per k the reference's `graph` is given
    kminmerData_abundance_prev.txt = its table of k - 1 and empty unitig files, and is ended once its tables are closed.  Only digests go
    into the repository (tests/golden/hifi_1m/manifest.json "multik": record count, sha256 of the sorted 20-byte records, the reference's
    checksum formula, the sum of abundances); tests/test_gpu_fullsize_multik.py checks the HIP path's tables against them at every k."""
    import dataclasses
    spec = synth.hifi_spec(1_000_000, seed=42, read_len=10_000, coverage=50.0)
    fasta = os.path.join(work, "hifi_1m.fasta")
    synth.write_fasta(fasta, spec)
    params = formats.Parameters(minimizer_size=15, kminmer_size=4, density=0.005, first_k=4, prev_k=4, last_k=last_k, hpc=True, data_type=0)
    tmp = run_ref_pipeline(os.path.join(work, "hifi_1m_multik"), fasta, params, threads=threads, graph=False)
    os.unlink(fasta)
    man_path = os.path.join(HERE, "hifi_1m", "manifest.json")
    manifest = json.load(open(man_path))
    cm, co = formats.parse_minimizer_reads(open(os.path.join(tmp, "read_data_corrected.txt"), "rb").read())
    assert formats.minimizer_reads_digest(cm, co) == manifest["read_data_corrected_digest"]        # the same reads as the k = 4 fixture
    per_k = {}
    prev_k = 4
    for k in range(4, last_k + 1):
        dataclasses.replace(params, kminmer_size=k, prev_k=prev_k).save(os.path.join(tmp, "parameters.gz"))
        if k > 4:
            shutil.copy(os.path.join(tmp, "kminmerData_abundance.txt"), os.path.join(tmp, "kminmerData_abundance_prev.txt"))
            for name in ("unitig_data.txt", "unitigGraph_prev.nodes.bin", "unitigGraph.nodes.refined_abundances.bin"):
                open(os.path.join(tmp, name), "wb").close()
        _graph_until_tables(tmp, threads, first_pass=(k == 4))
        raw = open(os.path.join(tmp, "kminmerData_abundance.txt"), "rb").read()
        rec = formats.parse_abundance_table(raw)
        with np.errstate(over="ignore"):
            checksum = int((rec["abundance"].astype(np.uint64) * rec["lo"]).sum(dtype=np.uint64))
        vec = open(os.path.join(tmp, "kminmerData_min.txt"), "rb").read() if k <= 5 else None
        per_k[str(k)] = dict(n_records=int(len(rec)), abundance_checksum=checksum, sum_abundance=int(rec["abundance"].astype(np.uint64).sum()),
                             **formats.table_digests(rec, vec, k))
        print(f"[make_golden] hifi_1m multik k={k}: {len(rec)} records", flush=True)
        prev_k = k
    assert per_k["4"]["abundance_sorted_sha256"] == manifest["abundance_sorted_sha256"]
    manifest["multik"] = dict(mode="benchmark mode (SURVEY.md 8(d)): reads only, previous table = the reference's own table of k - 1, empty unitig files; "
                                   "`graph` ended once its tables were closed", last_k=last_k, reference_threads=threads, per_k=per_k)
    with open(man_path, "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


_SHARED_GENOME = None


def _write_fastx_range(args) -> str:
    """reads [r0, r1) of `spec` to their own file (a worker process of make_ont_1m)"""
    path, spec, r0, r1 = args
    genome = _SHARED_GENOME          # made once by the parent, inherited through fork (400 MB; a copy per worker plus genome_codes' 64-bit
                                     # temporaries was 6 GB a worker)
    with open(path, "wb") as f:
        for a in range(r0, r1, 2000):
            b = min(a + 2000, r1)
            asc = synth.codes_to_ascii(synth.read_codes(spec, a, b, genome))
            qual = synth.read_qualities(spec, a, b)
            for j in range(b - a):
                f.write(b"@r%d\n" % (a + j)); f.write(asc[j].tobytes()); f.write(b"\n+\n"); f.write(qual[j].tobytes()); f.write(b"\n")
    return path


def make_ont_1m(work: str, threads: int = 8) -> None:
    """BASELINE.json configs[3]'s kind of input tied to the reference AT SIZE: 1,000,100 synthetic ONT reads x 20 kb with qualities
    (synth.ont_spec: 2 % errors, phred 10..39), ONE file -- so that the repetitive-minimizer census meets its cap of 1,000,001 reads per
    file (readSelection/ReadSelection.hpp:497-561, Commons.hpp:5873: the last 99 reads are scanned but not counted) -- through the
    reference's own `readSelection --skip-correction` and `graph --firstpass` (ended once its tables are closed).  40 GB of FASTQ and
    the products do not go into the repository: their digests do (tests/golden/ont_1m/manifest.json), with the few u32 the census
    chose (std::sort's order among equal counts at the cut is the one thing the reference leaves open: the test checks that the pick
    is A valid one against the device's counts, then scans with exactly it).  The reads are regenerated on the device from the seed."""
    import multiprocessing as mp
    global _SHARED_GENOME
    spec = synth.ont_spec(1_000_100, seed=42, read_len=20_000, coverage=50.0)
    g_len = int(spec.genome_offsets()[-1])
    _SHARED_GENOME = np.concatenate([synth.genome_codes(spec, a, min(a + (1 << 25), g_len)) for a in range(0, g_len, 1 << 25)])
    fastq = os.path.join(work, "ont_1m.fastq")
    cuts = [spec.n_reads * i // (4 * threads) for i in range(4 * threads + 1)]
    jobs = [(os.path.join(work, f"part{i:03d}.fastq"), spec, cuts[i], cuts[i + 1]) for i in range(4 * threads)]
    head = hashlib.sha256()
    with mp.get_context("fork").Pool(min(threads, 6)) as pool, open(fastq, "wb") as out:
        for i, part in enumerate(pool.imap(_write_fastx_range, jobs)):
            with open(part, "rb") as f:
                first = True
                for blk in iter(lambda: f.read(1 << 24), b""):
                    if i == 0 and first:
                        head.update(blk[: 1 << 20])
                    first = False
                    out.write(blk)
            os.unlink(part)
            print(f"[make_golden] ont_1m: part {i + 1} / {len(jobs)} written", flush=True)
    params = formats.Parameters(minimizer_size=15, kminmer_size=4, density=0.005, first_k=4, prev_k=4, last_k=0, hpc=False, data_type=1,
                                correction_density=0.025)
    tmp = run_ref_pipeline(os.path.join(work, "ont_1m"), fastq, params, threads=threads, extra_rs=["--skip-correction"], graph=False)
    fastq_bytes = os.path.getsize(fastq)
    os.unlink(fastq)
    print("[make_golden] ont_1m: readSelection done", flush=True)
    _graph_until_tables(tmp, threads, first_pass=True, timeout=4 * 3600)
    rec = formats.parse_abundance_table(open(os.path.join(tmp, "kminmerData_abundance.txt"), "rb").read())
    with np.errstate(over="ignore"):
        checksum = int((rec["abundance"].astype(np.uint64) * rec["lo"]).sum(dtype=np.uint64))
    cm, co = formats.parse_minimizer_reads(open(os.path.join(tmp, "read_data_corrected.txt"), "rb").read())
    rep = np.fromfile(os.path.join(tmp, "repetitiveMinimizers.bin"), "<u4")
    manifest = dict(
        kind="ont", config="BASELINE.json configs[3]'s input at 1,000,100 reads (one file: the census cap of 1,000,001 reads is crossed)",
        n_reads=spec.n_reads, read_len=spec.read_len, seed=spec.seed, coverage=50.0, sub_rate=spec.sub_rate, ins_rate=spec.ins_rate, del_rate=spec.del_rate,
        species_len=spec.species_len, species_weight=spec.species_weight, with_quality=True, fastq_bytes=fastq_bytes, fastq_first_mib_sha256=head.hexdigest(),
        K=15, density=0.005, correction_density=0.025, hpc=False, skip_correction=True, k=4, min_abundance=0, reference_threads=threads,
        repetitive_minimizers=[int(x) for x in rep],
        read_data_init_sha256=sha256(os.path.join(tmp, "read_data_init.txt")), read_data_init_bytes=os.path.getsize(os.path.join(tmp, "read_data_init.txt")),
        read_stats_hex=open(os.path.join(tmp, "read_stats.txt"), "rb").read().hex(),
        read_data_corrected_digest=formats.minimizer_reads_digest(cm, co), n_corrected_minimizers=int(len(cm)),
        n_records=int(len(rec)), abundance_checksum=checksum, sum_abundance=int(rec["abundance"].astype(np.uint64).sum()),
        reference_log=log_known_answers(tmp),
        **formats.table_digests(rec, open(os.path.join(tmp, "kminmerData_min.txt"), "rb").read(), 4))
    dst = os.path.join(HERE, "ont_1m")
    os.makedirs(dst, exist_ok=True)
    with open(os.path.join(dst, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("[make_golden] ont_1m: manifest written", flush=True)


def make_ont(work: str) -> None:
    # SURVEY 8(d) ONT R10 error model: 1 % substitutions + 0.5 % insertions + 0.5 % deletions, phred 10..39
    spec = synth.SynthSpec(n_reads=100, read_len=20_000, seed=11, sub_rate=0.01, ins_rate=0.005, del_rate=0.005,
                           species_len=[60_000, 50_000], species_weight=[0.7, 0.3], with_quality=True, name="ont")
    fastq = os.path.join(work, "ont_100.fastq")
    synth.write_fasta(fastq, spec)
    params = formats.Parameters(minimizer_size=15, kminmer_size=4, density=0.005, first_k=4, prev_k=4,
                                last_k=0, hpc=False, data_type=1, correction_density=0.025)
    tmp = run_ref_pipeline(os.path.join(work, "ont"), fastq, params, extra_rs=["--skip-correction"])
    store_outputs(tmp, os.path.join(HERE, "ont_100"), 4, dict(
        kind="ont", n_reads=spec.n_reads, read_len=spec.read_len, seed=spec.seed, sub_rate=spec.sub_rate,
        ins_rate=spec.ins_rate, del_rate=spec.del_rate,
        species_len=spec.species_len, species_weight=spec.species_weight, with_quality=True,
        fasta_sha256=sha256(fastq), K=15, density=0.005, hpc=False, k=4, min_abundance=0, skip_correction=True))


def make_rep(work: str) -> None:
    """SURVEY 8(a) A8, determineRepetitiveMinimizers (readSelection/ReadSelection.hpp:497-625) where its answer is
    unambiguous: enough distinct minimizers at the correction density for a cut of several (max(1, 1e-5 x distinct)), a small
    high-coverage species that supplies the top counts, and a seed for which no count ties across the cut -- std::sort's
    order among equal counts is the one thing the reference leaves open.  Only the manifest and the few u32 the reference
    chose are kept; the reads are regenerated from the seed."""
    from oracle import pyoracle as orc
    import ctypes as C
    L = orc.lib()
    L.orc_minimizer_parse.restype = C.c_size_t

    def census(spec):
        # with the oracle's parser (pinned above against the reference's MinimizerParser)
        genome = synth.genome_codes(spec)
        allm = []
        for r0 in range(0, spec.n_reads, 200):
            asc = synth.codes_to_ascii(synth.read_codes(spec, r0, min(r0 + 200, spec.n_reads), genome))
            for row in asc:
                sq = row.tobytes()
                n = len(sq)
                om = (C.c_uint32 * n)(); op = (C.c_uint32 * n)(); od = (C.c_uint8 * n)()
                k = L.orc_minimizer_parse(sq, C.c_size_t(n), 15, C.c_float(0.025), None, C.c_size_t(0), om, op, od)
                allm.append(np.frombuffer(om, np.uint32, k).copy())
        return np.unique(np.concatenate(allm), return_counts=True)

    for seed in range(5, 40):          # the first seed whose counts do not tie across the cut
        spec = synth.SynthSpec(n_reads=1600, read_len=20_000, seed=seed, sub_rate=0.01, ins_rate=0.005, del_rate=0.005,
                               species_len=[12_000_000, 30_000], species_weight=[0.85, 0.15], with_quality=True, name="ont")
        vals, counts = census(spec)
        n_keep = max(int(np.float32(0.00001) * np.float32(len(vals))), 1)
        order = np.sort(counts)[::-1]
        if n_keep >= 3 and order[n_keep - 1] > order[n_keep]:
            break
    else:
        raise RuntimeError("no seed without a tie across the cut")
    fastq = os.path.join(work, "ont_rep.fastq")
    synth.write_fasta(fastq, spec)
    params = formats.Parameters(minimizer_size=15, kminmer_size=4, density=0.005, first_k=4, prev_k=4,
                                last_k=0, hpc=False, data_type=1, correction_density=0.025)
    tmp = run_ref_pipeline(os.path.join(work, "ont_rep"), fastq, params, threads=8, extra_rs=["--skip-correction"], graph=False)
    rep = np.fromfile(os.path.join(tmp, "repetitiveMinimizers.bin"), "<u4")
    assert len(rep) == n_keep, (len(rep), n_keep)
    assert set(rep.tolist()) == set(vals[counts >= order[n_keep - 1]].tolist())
    dst = os.path.join(HERE, "ont_rep")
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(tmp, "repetitiveMinimizers.bin"), os.path.join(dst, "repetitiveMinimizers.bin"))
    with open(os.path.join(dst, "manifest.json"), "w") as f:
        json.dump(dict(kind="ont", n_reads=spec.n_reads, read_len=spec.read_len, seed=spec.seed, sub_rate=spec.sub_rate,
                       ins_rate=spec.ins_rate, del_rate=spec.del_rate, species_len=spec.species_len,
                       species_weight=spec.species_weight, with_quality=True, fasta_sha256=sha256(fastq), K=15,
                       density=0.005, correction_density=0.025, hpc=False, skip_correction=True,
                       n_distinct=int(len(vals)), n_keep=int(n_keep), cut_count=int(order[n_keep - 1]),
                       next_count=int(order[n_keep]), read_data_init_sha256=sha256(os.path.join(tmp, "read_data_init.txt"))),
                  f, indent=1, sort_keys=True)


MULTIK_INPUTS = ("parameters.gz", "kminmerData_abundance_prev.txt", "unitigGraph_prev.nodes.bin",
                 "unitigGraph.nodes.refined_abundances.bin", "unitig_data.txt")


def run_ref_multik(tmp: str, params: formats.Parameters, last_k: int, dst: str, manifest: dict, snapshot_ks=None) -> None:
    """The reference's own multi-k loop after readSelection (pipeline/AssemblyPipeline.hpp:609-671, executePass :1076-1145):
    per k write parameters.gz (k, prevK), run `graph`, then `contig` and `toMinspace`, which produce the next iteration's
    unitig_data.txt / *_prev files.  For every k > firstK the fixture keeps the INPUTS `graph` read (data files) and its
    hot-path OUTPUTS (sorted kminmerData_abundance.txt, kminmerData_min.txt at firstK+1, smallContigs_k<k>.bin)."""
    import dataclasses
    first_k = params.first_k
    prev_k = params.prev_k
    per_k = {}
    for k in range(first_k, last_k + 1):
        dataclasses.replace(params, kminmer_size=k, prev_k=prev_k, last_k=last_k).save(os.path.join(tmp, "parameters.gz"))
        cmd = [REFDRV, "graph", tmp, "--threads", "1"] + (["--min-abundance", "0", "--firstpass"] if k == first_k else [])
        log_path = os.path.join(os.path.dirname(tmp), "metaMDBG.log")
        log_start = os.path.getsize(log_path) if os.path.exists(log_path) else 0
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        if k > first_k and (snapshot_ks is None or k in snapshot_ks):
            d = os.path.join(dst, f"k{k}")
            os.makedirs(d, exist_ok=True)
            for name in MULTIK_INPUTS:
                shutil.copy(os.path.join(tmp, name), os.path.join(d, name))
            raw = open(os.path.join(tmp, "kminmerData_abundance.txt"), "rb").read()
            formats.sorted_abundance_records(raw).tofile(os.path.join(d, "kminmerData_abundance.sorted.bin"))
            if k == first_k + 1:
                raw = open(os.path.join(tmp, "kminmerData_min.txt"), "rb").read()
                formats.sorted_vector_records(raw, k).astype("<u4").tofile(os.path.join(d, "kminmerData_min.sorted.bin"))
            shutil.copy(os.path.join(tmp, "smallContigs", f"smallContigs_k{k}.bin"), os.path.join(d, "smallContigs.bin"))
            known = log_known_answers(tmp, log_start)       # EdgeIndexer / UnitigEdgeIndexer run while vectors exist (k <= firstK+1)
            if "n_unitig_edges" in known:
                shutil.copy(os.path.join(tmp, "unitigGraph.nodes.bin"), os.path.join(d, "unitigGraph.nodes.bin"))
            per_k[str(k)] = dict(reference_log=known, n_records=len(raw) // 20 if k > first_k + 1 else os.path.getsize(os.path.join(tmp, "kminmerData_abundance.txt")) // 20,
                                 small_contigs_bytes=os.path.getsize(os.path.join(d, "smallContigs.bin")))
        if k == last_k:
            break
        subprocess.run([REFDRV, "contig", tmp, "--threads", "1", "--max-bubble-length", "50000", "--max-tip-length", "50000"],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        subprocess.run([REFDRV, "toMinspace", tmp, os.path.join(tmp, "contigs.nodepath"), os.path.join(tmp, "unitig_data.txt"),
                        os.path.join(tmp, "unitigGraph.nodes.bin"), "--threads", "1"],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        prev_k = k
    shutil.copy(os.path.join(tmp, "read_data_corrected.txt"), os.path.join(dst, "read_data_corrected.txt"))
    with open(os.path.join(dst, "manifest.json"), "w") as f:
        json.dump(dict(manifest, first_k=first_k, last_k=last_k, per_k=per_k, **({"ks": sorted(int(x) for x in per_k)} if snapshot_ks is not None else {})),
                  f, indent=1, sort_keys=True)


def make_multik(work: str, only_ont: bool = False) -> None:
    """hifi (HPC, 300 reads, three species) and ont (no HPC, 2 % errors, --skip-correction) through k = 4..11."""
    if not only_ont:
        _make_multik_hifi(work)
    _make_multik_ont(work)


def _make_multik_hifi(work: str) -> None:
    spec = synth.hifi_spec(300, seed=19, coverage=30.0)
    fasta = os.path.join(work, "hifi_multik.fasta")
    synth.write_fasta(fasta, spec)
    params = formats.Parameters(minimizer_size=15, kminmer_size=4, density=0.005, first_k=4, prev_k=4,
                                last_k=0, hpc=True, data_type=0)
    tmp = run_ref_pipeline(os.path.join(work, "hifi_multik"), fasta, params, graph=False)
    run_ref_multik(tmp, params, 11, os.path.join(HERE, "hifi_multik"), dict(
        kind="hifi", n_reads=spec.n_reads, read_len=spec.read_len, seed=spec.seed, sub_rate=spec.sub_rate,
        species_len=spec.species_len, species_weight=spec.species_weight, fasta_sha256=sha256(fasta), K=15, density=0.005, hpc=True))


def _make_multik_ont(work: str) -> None:
    spec = synth.SynthSpec(n_reads=150, read_len=20_000, seed=23, sub_rate=0.01, ins_rate=0.005, del_rate=0.005,
                           species_len=[50_000, 40_000], species_weight=[0.7, 0.3], with_quality=True, name="ont")
    fastq = os.path.join(work, "ont_multik.fastq")
    synth.write_fasta(fastq, spec)
    params = formats.Parameters(minimizer_size=15, kminmer_size=4, density=0.005, first_k=4, prev_k=4,
                                last_k=0, hpc=False, data_type=1, correction_density=0.025)
    tmp = run_ref_pipeline(os.path.join(work, "ont_multik"), fastq, params, extra_rs=["--skip-correction"], graph=False)
    run_ref_multik(tmp, params, 11, os.path.join(HERE, "ont_multik"), dict(
        kind="ont", n_reads=spec.n_reads, read_len=spec.read_len, seed=spec.seed, sub_rate=spec.sub_rate,
        ins_rate=spec.ins_rate, del_rate=spec.del_rate,
        species_len=spec.species_len, species_weight=spec.species_weight, with_quality=True, fasta_sha256=sha256(fastq),
        K=15, density=0.005, hpc=False, skip_correction=True))


DEEP_KS = (12, 13, 14, 15, 16, 24, 31, 32, 33, 48, 64, 65)


def make_deepk(work: str, which: str = "both") -> None:
    """The range of k the reference's DEFAULT `asm` runs: no --max-k, lastK = N50 x density x 2 (Commons.hpp:1726-1741; loop step 1,
    :1986-1987, pipeline/AssemblyPipeline.hpp:603-671) -- k = 4 .. 100 for 10 kb HiFi reads, 4 .. 200 for 20 kb ONT reads.  The whole
    loop is run by the reference's own code (graph -> contig -> toMinspace per k); the files `graph` read and the tables it wrote are
    kept at k in DEEP_KS and at lastK: the generic window hash (odd tails of 1 / 3 words at k = 13, 15, 33), k above the partitioned
    pass's 32, k above most reads' minimizer count (37 for these HiFi reads: from there on the unitigs carry the table)."""
    if which in ("both", "hifi"):
        spec = synth.hifi_spec(1000, seed=29, coverage=40.0)
        fasta = os.path.join(work, "hifi_deepk.fasta")
        synth.write_fasta(fasta, spec)
        params = formats.Parameters(minimizer_size=15, kminmer_size=4, density=0.005, first_k=4, prev_k=4, last_k=0, hpc=True, data_type=0)
        tmp = run_ref_pipeline(os.path.join(work, "hifi_deepk"), fasta, params, graph=False)
        stats = formats.parse_read_stats(open(os.path.join(tmp, "read_stats.txt"), "rb").read())
        last_k = int(subprocess.run([REFDRV, "fn_lastk", "0.005", str(stats["n50"]), "4", "0"], capture_output=True, text=True, check=True).stdout)
        run_ref_multik(tmp, params, last_k, os.path.join(HERE, "hifi_deepk"), dict(
            kind="hifi", n_reads=spec.n_reads, read_len=spec.read_len, seed=spec.seed, sub_rate=spec.sub_rate, n50=int(stats["n50"]),
            species_len=spec.species_len, species_weight=spec.species_weight, fasta_sha256=sha256(fasta), K=15, density=0.005, hpc=True),
            snapshot_ks=set(DEEP_KS) | {last_k})
    if which in ("both", "ont"):
        spec = synth.SynthSpec(n_reads=150, read_len=20_000, seed=23, sub_rate=0.01, ins_rate=0.005, del_rate=0.005,
                               species_len=[50_000, 40_000], species_weight=[0.7, 0.3], with_quality=True, name="ont")
        fastq = os.path.join(work, "ont_deepk.fastq")
        synth.write_fasta(fastq, spec)
        params = formats.Parameters(minimizer_size=15, kminmer_size=4, density=0.005, first_k=4, prev_k=4, last_k=0, hpc=False, data_type=1,
                                    correction_density=0.025)
        tmp = run_ref_pipeline(os.path.join(work, "ont_deepk"), fastq, params, extra_rs=["--skip-correction"], graph=False)
        stats = formats.parse_read_stats(open(os.path.join(tmp, "read_stats.txt"), "rb").read())
        last_k = int(subprocess.run([REFDRV, "fn_lastk", "0.005", str(stats["n50"]), "4", "0"], capture_output=True, text=True, check=True).stdout)
        run_ref_multik(tmp, params, last_k, os.path.join(HERE, "ont_deepk"), dict(
            kind="ont", n_reads=spec.n_reads, read_len=spec.read_len, seed=spec.seed, sub_rate=spec.sub_rate, n50=int(stats["n50"]),
            ins_rate=spec.ins_rate, del_rate=spec.del_rate, species_len=spec.species_len, species_weight=spec.species_weight,
            with_quality=True, fasta_sha256=sha256(fastq), K=15, density=0.005, hpc=False, skip_correction=True),
            snapshot_ks=set(DEEP_KS) | {100, last_k})


def edge_reads() -> list[bytes]:
    rng = np.random.default_rng(1234)

    def rnd(n):
        return bytes(synth.CODE2ASCII[rng.integers(0, 4, n)])
    reads = [
        rnd(3000),                                        # plain
        # (reads containing N/n are NOT here: the reference's computeSequenceComplexity indexes
        #  kmerCounts[-1] for them -- readSelection/ReadSelection.hpp:1196 -- and aborts with heap
        #  corruption; N handling is pinned at function level by fn_scan below)
        rnd(2000).lower(),                                # lowercase (same 2-bit code, different HPC chars)
        rnd(14),                                          # shorter than K
        rnd(15), rnd(16), rnd(17),                        # 1, 2, 3 k-mers -> loop [1, nK-1)
        rnd(40),                                          # < 66: no full complexity window -> NaN
        b"ACGT" * 700,                                    # periodic
        b"A" * 500 + rnd(1000) + b"C" * 800 + rnd(1200),  # long homopolymers
        b"AC" * 1500,                                     # low complexity (score > 5)
        b"AAT" * 1200,                                    # low complexity
        rnd(5000)[:4999],
        b"A" * 64,                                        # HPC collapses to one base
        rnd(1000) + b"AAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA",  # ends in a run
        b"TTTTTTTTTTTTTTTTTTTTTTTTTTTT" + rnd(1000),      # starts with a run
    ]
    # reverse-complement pair -> same canonical minimizers, opposite directions
    a = rnd(4000)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    reads += [a, a.translate(comp)[::-1]]
    return reads


def make_edge(work: str) -> None:
    reads = edge_reads()
    dst = os.path.join(HERE, "edge")
    os.makedirs(dst, exist_ok=True)
    fasta = os.path.join(dst, "edge.fasta")
    with open(fasta, "wb") as f:
        for i, s in enumerate(reads):
            f.write(b">e%d\n" % i + s + b"\n")
    for tag, K, hpc in (("hpc_k15", 15, True), ("nohpc_k15", 15, False), ("hpc_k16", 16, True), ("nohpc_k13", 13, False)):
        params = formats.Parameters(minimizer_size=K, kminmer_size=4, density=0.005 if K != 13 else 0.02,
                                    first_k=4, prev_k=4, hpc=hpc, data_type=0 if hpc else 1)
        tmp = run_ref_pipeline(os.path.join(work, "edge_" + tag), fasta, params,
                               extra_rs=[] if hpc else ["--skip-correction"], graph=False)
        shutil.copy(os.path.join(tmp, "read_data_init.txt"), os.path.join(dst, f"read_data_init.{tag}.txt"))
        shutil.copy(os.path.join(tmp, "read_stats.txt"), os.path.join(dst, f"read_stats.{tag}.txt"))
        shutil.copy(os.path.join(tmp, "repetitiveMinimizers.bin"), os.path.join(dst, f"repetitiveMinimizers.{tag}.bin"))
    # FASTQ variant of the same reads (qualities exercise getMinQuality through rlePositions)
    rng = np.random.default_rng(99)
    fastq = os.path.join(dst, "edge.fastq")
    with open(fastq, "wb") as f:
        for i, s in enumerate(reads):
            q = bytes((rng.integers(0, 60, len(s)) + 33).astype(np.uint8))
            f.write(b"@e%d\n" % i + s + b"\n+\n" + q + b"\n")
    for tag, hpc in (("fastq_hpc_k15", True), ("fastq_nohpc_k15", False)):
        params = formats.Parameters(minimizer_size=15, kminmer_size=4, density=0.005, first_k=4, prev_k=4,
                                    hpc=hpc, data_type=0 if hpc else 1)
        tmp = run_ref_pipeline(os.path.join(work, "edge_" + tag), fastq, params,
                               extra_rs=[] if hpc else ["--skip-correction"], graph=False)
        shutil.copy(os.path.join(tmp, "read_data_init.txt"), os.path.join(dst, f"read_data_init.{tag}.txt"))
        shutil.copy(os.path.join(tmp, "repetitiveMinimizers.bin"), os.path.join(dst, f"repetitiveMinimizers.{tag}.bin"))


def refdrv_lines(args: list[str], lines: list[str]) -> list[str]:
    r = subprocess.run([REFDRV] + args, input="\n".join(lines) + "\n", capture_output=True, text=True, check=True)
    return r.stdout.strip("\n").split("\n")


def make_fn() -> None:
    dst = os.path.join(HERE, "fn")
    os.makedirs(dst, exist_ok=True)
    rng = np.random.default_rng(2024)
    out = {}
    # murmur
    vals = [0, 1, 12345, 2**30 - 1, 2**32 - 1, 2**64 - 1] + [int(x) for x in rng.integers(0, 2**32, 64)]
    out["murmur"] = dict(inputs=[str(v) for v in vals], outputs=refdrv_lines(["fn_murmur"], [str(v) for v in vals]))
    # kminmer normalize + hash128 for k = 2..13 (all Murmur tail lengths), incl. palindromes/ties
    km = {}
    for k in range(2, 14):
        vecs = [rng.integers(0, 2**32, k).tolist() for _ in range(8)]
        half = rng.integers(0, 2**32, (k + 1) // 2).tolist()
        vecs.append(half + half[: k // 2][::-1])                 # palindrome
        v = rng.integers(0, 2**32, k).tolist(); v[-1] = v[0]      # tie on first compare
        vecs.append(v)
        vecs.append(sorted(rng.integers(0, 1000, k).tolist(), reverse=True))
        lines = [" ".join(map(str, v)) for v in vecs]
        km[str(k)] = dict(inputs=lines, outputs=refdrv_lines(["fn_kminmer", str(k)], lines))
    out["kminmer"] = km
    # purgePalindrome: crafted lists with palindromic windows, nested, overlapping
    lists = [
        [1, 2, 3, 4, 5, 6, 7, 8],
        [1, 2, 2, 1, 5, 6, 7, 8],
        [9, 1, 2, 3, 2, 1, 7, 8, 4],
        [1, 2, 3, 3, 2, 1, 1, 2, 3, 3, 2, 1],
        [5, 5, 5, 5, 5, 5, 5],
        [1, 2, 1, 2, 1, 2, 1, 2, 1],
        [7, 1, 2, 3, 4, 4, 3, 2, 1, 8, 9, 10],
        [1, 2, 3],
        [],
        [4, 4],
        [3, 1, 4, 1, 5, 9, 2, 6, 5, 3, 5, 8, 9, 7, 9, 3, 2, 3, 8, 4, 6, 2, 6, 4, 3, 3, 8, 3, 2, 7, 9, 5],
    ]
    for _ in range(40):
        n = int(rng.integers(4, 40))
        lists.append(rng.integers(0, 4, n).tolist())          # tiny alphabet -> many palindromes
    for _ in range(20):
        n = int(rng.integers(4, 60))
        lists.append(rng.integers(0, 12, n).tolist())
    lines = [" ".join(map(str, v)) for v in lists]
    out["purge"] = {}
    for fk, lk in ((4, 100), (4, 6), (3, 9)):
        out["purge"][f"{fk}_{lk}"] = dict(inputs=lines, outputs=refdrv_lines(["fn_purge", str(fk), str(lk)], lines))
    # MinimizerParser on reads with invalid characters (N/n), both HPC settings
    def rnd(n):
        return bytes(synth.CODE2ASCII[rng.integers(0, 4, n)]).decode()
    nreads = [rnd(1500) + "N" + rnd(1500), rnd(700) + "NNNNNNNNNN" + rnd(900) + "n" + rnd(400),
              "N" + rnd(600), rnd(600) + "N", rnd(10) + "N" + rnd(10), "N" * 40, rnd(2000),
              rnd(300) + "NN" + rnd(20) + "N" + rnd(300), rnd(800).lower() + "N" + rnd(800)]
    out["scan_n"] = {}
    for hpc in (0, 1):
        for K, dens in ((15, 0.02), (16, 0.02), (11, 0.05)):
            out["scan_n"][f"K{K}_hpc{hpc}"] = dict(
                K=K, density=dens, hpc=hpc, inputs=nreads,
                outputs=refdrv_lines(["fn_scan", str(K), str(dens), str(hpc)], nreads))
    # lastK
    cases = [(0.005, 10000, 4, 0), (0.005, 10000, 4, 11), (0.005, 20000, 4, 0), (0.005, 300, 4, 0),
             (0.025, 9000, 4, 0), (0.005, 15431, 4, 0), (0.005, 100, 4, 0)]
    out["lastk"] = [dict(args=list(c), out=int(subprocess.run(
        [REFDRV, "fn_lastk"] + [str(x) for x in c], capture_output=True, text=True, check=True).stdout)) for c in cases]
    # Utils::applyDensityThreshold: kept indices of random minimizer lists (own generator: the cases above stay as they were)
    rng2 = np.random.default_rng(20260926)
    dl = [" ".join(map(str, rng2.integers(0, 2**32, int(n)).tolist())) for n in (0, 1, 7, 400, 3000)]
    out["density"] = {str(d): dict(inputs=dl, outputs=refdrv_lines(["fn_density", str(d)], dl)) for d in (0.005, 0.025, 0.2, 0.5)}
    # the correction scan (ReadCorrection::ReadSelectionFunctor): minimizers + min quality over [rle[pos], rle[pos+l-1]]
    def rnd_hp(n):   # homopolymer-rich
        base = synth.CODE2ASCII[rng2.integers(0, 4, n)]
        return bytes(np.repeat(base, rng2.choice([1, 1, 1, 2, 3, 6], n))).decode()
    creads = [rnd_hp(n) for n in (40, 500, 2500, 2100, 5000)] + ["ACGT" * 300, "A" * 50 + rnd_hp(200) + "T" * 70]
    cquals = ["".join(chr(33 + int(q)) for q in rng2.integers(0, 60, len(r))) for r in creads]
    clines = [f"{r} {q}" for r, q in zip(creads, cquals)]
    out["corrscan"] = {}
    for hpc in (0, 1):
        for K, dens in ((13, 0.025), (15, 0.025), (16, 0.05)):
            out["corrscan"][f"K{K}_hpc{hpc}"] = dict(K=K, density=dens, hpc=hpc, reads=creads, quals=cquals,
                                                     outputs=refdrv_lines(["fn_corrscan", str(K), str(dens), str(hpc)], clines))
    # MinimizerParser with _trimBps = 0 (GenerateGfa's unitig scan): high density so the end l-mers do get selected
    treads = [rnd_hp(n) for n in (15, 16, 17, 18, 40, 300, 2048 * 2 + 15, 2048 + 16, 5000)] + ["ACGTTGCA" * 40]
    out["scan_notrim"] = {}
    for hpc in (0, 1):
        for K, dens in ((15, 0.5), (16, 0.3), (11, 0.9)):
            out["scan_notrim"][f"K{K}_hpc{hpc}"] = dict(K=K, density=dens, hpc=hpc, inputs=treads,
                                                        outputs=refdrv_lines(["fn_scan_notrim", str(K), str(dens), str(hpc)], treads))
    # EncoderRLE compares CHARACTERS (Commons.hpp:4177-4178) while k-mers see 2-bit codes (utils/kmer/Kmer.hpp:462): mixed
    # case ("aA" is two runs), soft-masked blocks, IUPAC letters that share a code with a neighbour ("CR", "GK"), case flips
    # inside homopolymers across the 32-base word and 2048-base tile borders of the device layout
    rng3 = np.random.default_rng(20260927)

    def rnd3(n):
        return bytes(synth.CODE2ASCII[rng3.integers(0, 4, n)]).decode()

    def soft_masked(n):
        s = bytearray(rnd_hp3(n).encode())
        for _ in range(max(1, n // 400)):
            a = int(rng3.integers(0, len(s))); b = min(len(s), a + int(rng3.integers(1, 200)))
            s[a:b] = bytes(s[a:b]).lower()
        return s.decode()

    def rnd_hp3(n):
        base = synth.CODE2ASCII[rng3.integers(0, 4, n)]
        return bytes(np.repeat(base, rng3.choice([1, 1, 1, 2, 3, 6], n))).decode()

    def flip_each(n):        # every base upper or lower at random: many "aA" pairs
        s = bytearray(rnd_hp3(n).encode())
        m = rng3.integers(0, 2, len(s)).astype(bool)
        return bytes(c | 0x20 if f else c for c, f in zip(s, m)).decode()

    def iupac(n):            # letters without bit 3 that share a 2-bit code with a base: R,S (C), U,T (T), W,V (G), B,Q (C/A)
        s = bytearray(rnd_hp3(n).encode())
        for _ in range(max(2, n // 100)):
            s[int(rng3.integers(0, len(s)))] = ord(rng3.choice(list("RSUWVBQacgt")))
        return s.decode()

    borders = "A" * 31 + "a" * 3 + rnd3(2048 - 34 - 5) + "G" * 5 + "g" * 4 + rnd3(300) + "TtTtTTtt" + rnd3(100)
    cases = [soft_masked(3000), soft_masked(700), flip_each(2500), flip_each(40), iupac(2600), iupac(300), borders,
             "aAaAaAaA" * 10 + rnd3(200), rnd3(500).lower(), "cCGg" + rnd3(30)]
    out["scan_case"] = {}
    for hpc in (0, 1):
        for K, dens in ((15, 0.05), (16, 0.05), (11, 0.1)):
            out["scan_case"][f"K{K}_hpc{hpc}"] = dict(K=K, density=dens, hpc=hpc, inputs=cases,
                                                      outputs=refdrv_lines(["fn_scan", str(K), str(dens), str(hpc)], cases))
    # _trimBps = 0 at the device kernel's block borders: compressed lengths 2048*b + K + {-1, 0, 1}, at a
    # density low enough for the reads to stay in the block kernel's row stage (own generator: the cases above stay as they were)
    rng4 = np.random.default_rng(20260928)

    def no_runs(n):          # n bases, no two neighbours equal
        c = np.empty(n, dtype=np.int64)
        c[0] = rng4.integers(0, 4)
        c[1:] = rng4.integers(1, 4, n - 1)
        return np.cumsum(c) % 4

    def with_compressed_length(n, hpc):
        codes = no_runs(n)
        if hpc:
            codes = np.repeat(codes, rng4.choice([1, 1, 1, 2, 3], n))
        return bytes(synth.CODE2ASCII[codes]).decode()

    out["scan_notrim_block"] = {}
    for hpc in (0, 1):
        for K, dens in ((15, 0.05), (16, 0.05), (11, 0.04)):
            breads = [with_compressed_length(2048 * b + K + d, hpc) for b in (1, 2, 3) for d in (-1, 0, 1)]
            out["scan_notrim_block"][f"K{K}_hpc{hpc}"] = dict(
                K=K, density=dens, hpc=hpc, inputs=breads,
                outputs=refdrv_lines(["fn_scan_notrim", str(K), str(dens), str(hpc)], breads))
    with open(os.path.join(dst, "fn_golden.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)


def main() -> None:
    if not os.path.exists(REFDRV):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
    work = tempfile.mkdtemp(prefix="mdbg_golden_", dir=os.environ.get("MDBG_GOLDEN_SCRATCH"))
    try:
        if "--only-1m" not in sys.argv:
            make_fn()
        if "--only-fn" in sys.argv:
            return
        if "--only-1m" in sys.argv:      # minutes of CPU and 10 GB of scratch: not part of the default regeneration
            if "--multik" in sys.argv:   # ... and the loop k = 4 .. 11 on the same reads (benchmark mode), digests into the same manifest
                # (--last-k N: further, e.g. 24 -- the generic window hash behind k >= 12 against the reference at a million reads)
                make_hifi_1m_multik(work, last_k=int(sys.argv[sys.argv.index("--last-k") + 1]) if "--last-k" in sys.argv else 11)
            else:
                make_hifi_1m(work)
            return
        if "--only-ont-1m" in sys.argv:  # about half an hour of CPU and 80 GB of scratch
            make_ont_1m(work)
            return
        if "--deep-k" in sys.argv:       # the reference's loop to lastK(N50): a few minutes of CPU
            make_deepk(work, "hifi" if "--hifi" in sys.argv else "ont" if "--ont" in sys.argv else "both")
            return
        if "--only-multik" in sys.argv:
            make_multik(work)
            return
        if "--only-rep" in sys.argv:
            make_rep(work)
            return
        if "--only-ont" in sys.argv:
            make_ont(work)
            make_multik(work, only_ont=True)
            make_rep(work)
            return
        if "--only-pipelines" in sys.argv:
            make_hifi(work)
            make_ont(work)
            return
        make_edge(work)
        make_hifi(work)
        make_ont(work)
        make_multik(work)
        make_rep(work)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    print("golden fixtures written under", HERE)


if __name__ == "__main__":
    main()
