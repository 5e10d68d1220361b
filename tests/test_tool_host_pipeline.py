"""The host pipeline of the C++ drop-in (metamdbg_amd/host/mdbg_tool.cpp readSelection: parser workers -> consumers one batch ahead ->
record builders writing in place -> the statistics thread in read order -> the purge pass) driven on a CPU through thousands of
batches against a TEST DOUBLE of the library (tests/host/stub_mdbg_hip.cpp: fake minimizers that depend on the read's length only).
No GPU, no numerics -- order, completeness and freedom from deadlock, under several batch sizes, thread counts and timings: the output
must be the same bytes every time, and exactly what the fake minimizers imply.  (A missed wake-up between the consumers and the record
builders once hung the tool on a 50 Gbp file and took half an hour of GPU time to find out; this test reproduces it in a second.)"""
from __future__ import annotations

import os
import re
import struct
import subprocess

import numpy as np
import pytest

from metamdbg_amd import formats

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def stub_tool(tmp_path_factory):
    d = tmp_path_factory.mktemp("stubtool")
    lib = str(d / "libmdbg_hip.so")
    exe = str(d / "mdbg_tool")
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", os.path.join(ROOT, "tests", "host", "stub_mdbg_hip.cpp"), "-o", lib, "-lpthread"], check=True)
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "metamdbg_amd", "host", "mdbg_tool.cpp"), "-o", exe, "-L" + str(d), "-lmdbg_hip",
                    "-lz", "-lpthread", "-Wl,-rpath," + str(d)], check=True)
    return exe


@pytest.fixture(scope="module")
def read_set(tmp_path_factory):
    d = tmp_path_factory.mktemp("stubreads")
    rng = np.random.default_rng(5)
    lens = rng.integers(0, 4000, 30_000).astype(np.int64)
    lens[::97] = 0
    pool = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 200_000)]
    fasta = str(d / "reads.fasta")
    with open(fasta, "wb") as f:
        for i, n in enumerate(lens):
            a = int(rng.integers(0, len(pool) - 4000))
            f.write(b">r%d\n" % i + pool[a:a + int(n)].tobytes() + b"\n")
    # what the stub's fake minimizers imply for read_data_init.txt / read_data_corrected.txt
    init, corr = [], []
    for n in lens:
        L, k = int(n), int(n) // 271
        m = ((L * 2654435761 + np.arange(k, dtype=np.uint64) * 40503) & 0xFFFFFFFF).astype("<u4")
        init.append(struct.pack("<IB", k, 0) + m.tobytes() + (np.arange(k, dtype="<u4") * 271).astype("<u4").tobytes() +
                    ((L + np.arange(k)) & 1).astype("u1").tobytes() + np.ones(k, "u1").tobytes() + struct.pack("<II", 0xFFC00000, L))
        corr.append(struct.pack("<IB", k, 0) + m.tobytes())
    return fasta, b"".join(init), b"".join(corr), lens


@pytest.mark.parametrize("batch_bases,threads,jitter,ont,gpus,consumers", [
    (1 << 20, 32, 0, False, 1, 0), (1 << 20, 32, 30, False, 1, 0), (1 << 16, 8, 0, False, 1, 0), (1 << 18, 3, 100, False, 1, 0),
    (1 << 25, 16, 0, False, 1, 0), (1 << 14, 32, 0, False, 1, 0), (1 << 18, 16, 20, True, 1, 0), (1 << 15, 32, 0, True, 1, 0),
    # more than two consumers (round-3 ADVICE: both hung): --gpus 2 / 4 / 8 (two consumers per device from 8 threads up), MDBG_TOOL_CONSUMERS
    (1 << 15, 8, 0, False, 2, 0), (1 << 15, 8, 20, False, 6, 0), (1 << 15, 16, 0, False, 8, 0), (1 << 14, 8, 0, False, 8, 0),
    (1 << 15, 1, 0, False, 4, 0), (1 << 15, 10, 10, False, 8, 0), (1 << 15, 16, 0, False, 1, 3), (1 << 15, 16, 30, False, 1, 4),
    (1 << 15, 16, 10, True, 4, 0)])
def test_read_selection_pipeline_is_ordered_complete_and_does_not_hang(stub_tool, read_set, tmp_path, batch_bases, threads, jitter, ont, gpus, consumers):
    """ont: the ONT preset -- the census pre-pass (one batch ahead, ReadSelection.hpp:497-561) in front of the main pass, --skip-correction.
    gpus / consumers: several consumer threads (the test double has as many devices as it is asked for)."""
    fasta, exp_init, exp_corr, lens = read_set
    tmp = tmp_path / "out" / "tmp"
    os.makedirs(tmp / "filter")
    formats.Parameters(minimizer_size=15, kminmer_size=4, density=0.005, first_k=4, prev_k=4, hpc=not ont, data_type=1 if ont else 0,
                       correction_density=0.025).save(str(tmp / "parameters.gz"))
    (tmp / "input.txt").write_text(fasta + "\n")
    env = dict(os.environ, MDBG_STUB_JITTER_US=str(jitter), MDBG_TRACE="1")
    if consumers:
        env["MDBG_TOOL_CONSUMERS"] = str(consumers)
    for rep in range(3):
        r = subprocess.run([stub_tool, "readSelection", str(tmp), str(tmp / "read_data_init.txt"), str(tmp / "input.txt"), "--threads", str(threads),
                            "--min-read-quality", "0.000000", "--batch-bases", str(batch_bases), "--gpus", str(gpus)] + (["--skip-correction"] if ont else []),
                           env=env, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-800:]
        assert (tmp / "read_data_init.txt").read_bytes() == exp_init
        assert (tmp / "read_data_corrected.txt").read_bytes() == exp_corr
        # the purge pass's group slabs: allocated beside the main pass when the input is many groups long, never more than wanted
        m = re.search(r"(\d+) of (\d+) group slabs were allocated beside the main pass", r.stderr)
        assert m, r.stderr[-800:]
        assert int(m.group(1)) <= int(m.group(2))
        assert (int(m.group(2)) > 0) == (os.path.getsize(fasta) // (batch_bases * 32) >= 2)
        st = formats.parse_read_stats((tmp / "read_stats.txt").read_bytes())
        assert st["n_reads"] == len(lens) and st["n_bases"] == int(lens.sum()) and st["n_minimizers"] == int((lens // 271).sum())


def test_pipeline_under_thread_sanitizer(read_set, tmp_path):
    """The same pipeline built with -fsanitize=thread (tool and test double): no data race between the feeder's workers, the consumers, the
    record builders, the statistics thread and the purge pass, in the HiFi and the ONT shape of the run."""
    fasta, exp_init, exp_corr, lens = read_set
    d = tmp_path / "tsan"
    os.makedirs(d)
    lib, exe = str(d / "libmdbg_hip.so"), str(d / "mdbg_tool")
    flags = ["-O1", "-g", "-fsanitize=thread", "-std=c++17"]
    r = subprocess.run(["g++"] + flags + ["-shared", "-fPIC", os.path.join(ROOT, "tests", "host", "stub_mdbg_hip.cpp"), "-o", lib, "-lpthread"], capture_output=True, text=True)
    if r.returncode == 0:
        r = subprocess.run(["g++"] + flags + [os.path.join(ROOT, "metamdbg_amd", "host", "mdbg_tool.cpp"), "-o", exe, "-L" + str(d), "-lmdbg_hip", "-lz", "-lpthread",
                            "-Wl,-rpath," + str(d)], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("no ThreadSanitizer in this toolchain: " + r.stderr[-200:])
    for ont in (False, True):
        tmp = tmp_path / ("ont" if ont else "hifi") / "tmp"
        os.makedirs(tmp / "filter")
        formats.Parameters(minimizer_size=15, kminmer_size=4, density=0.005, first_k=4, prev_k=4, hpc=not ont, data_type=1 if ont else 0,
                           correction_density=0.025).save(str(tmp / "parameters.gz"))
        (tmp / "input.txt").write_text(fasta + "\n")
        r = subprocess.run([exe, "readSelection", str(tmp), str(tmp / "read_data_init.txt"), str(tmp / "input.txt"), "--threads", "16", "--min-read-quality", "0.000000",
                            "--batch-bases", str(1 << 17)] + (["--skip-correction"] if ont else []),
                           env=dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66"), capture_output=True, text=True, timeout=300)
        assert "ThreadSanitizer" not in r.stderr and r.returncode == 0, r.stderr[-1500:]
        assert (tmp / "read_data_init.txt").read_bytes() == exp_init and (tmp / "read_data_corrected.txt").read_bytes() == exp_corr


@pytest.mark.parametrize("batch_bases,threads,jitter,ont", [(1 << 18, 16, 0, False), (1 << 15, 8, 30, False), (1 << 16, 3, 0, False), (1 << 17, 16, 10, True), (1 << 25, 4, 0, False)])
def test_asm_step_is_the_two_commands_in_one_process(stub_tool, read_set, tmp_path, batch_bases, threads, jitter, ont):
    """`mdbg_tool asmStep` = readSelection + graph --firstpass in ONE process: the second command finds the library context alive and the
    corrected minimizers still on the device -- no second context, no read_data_corrected.txt parsed back (pipeline/AssemblyPipeline.hpp:716-740,
    :763-792 run them as two children).  Every file must be what the two commands write: read_data_init.txt, read_stats.txt and
    read_data_corrected.txt byte for byte, the tables as multisets (the first pass is handed the purged groups in the order they were purged
    in; the test double's rows are a function of the reads alone), the log lines of both, one perf.bin."""
    fasta, exp_init, exp_corr, lens = read_set
    P = formats.Parameters(minimizer_size=15, kminmer_size=4, density=0.005, first_k=4, prev_k=4, hpc=not ont, data_type=1 if ont else 0, correction_density=0.025)
    env = dict(os.environ, MDBG_STUB_JITTER_US=str(jitter))
    out = {}
    for mode in ("two", "one"):
        tmp = tmp_path / mode / "tmp"
        for d in ("filter", "smallContigs"):
            os.makedirs(tmp / d)
        P.save(str(tmp / "parameters.gz"))
        (tmp / "input.txt").write_text(fasta + "\n")
        rs = [str(tmp), str(tmp / "read_data_init.txt"), str(tmp / "input.txt"), "--threads", str(threads), "--min-read-quality", "0.000000",
              "--batch-bases", str(batch_bases)] + (["--skip-correction"] if ont else [])
        cmds = [[stub_tool, "readSelection"] + rs, [stub_tool, "graph", str(tmp), "--threads", str(threads), "--min-abundance", "0", "--firstpass"]] if mode == "two" \
            else [[stub_tool, "asmStep"] + rs + ["--min-abundance", "0"]]
        for cmd in cmds:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=120)
            assert r.returncode == 0, r.stderr[-800:]
        out[mode] = {n: (tmp / n).read_bytes() for n in ("read_data_init.txt", "read_data_corrected.txt", "read_stats.txt", "kminmerData_abundance.txt",
                                                          "kminmerData_abundance_init.txt", "kminmerData_min.txt", "perf.bin", "smallContigs/smallContigs_k4.bin")}
        out[mode]["log"] = (tmp_path / mode / "metaMDBG.log").read_text()
    one, two = out["one"], out["two"]
    assert one["read_data_init.txt"] == two["read_data_init.txt"] == exp_init
    assert one["read_data_corrected.txt"] == two["read_data_corrected.txt"] == exp_corr
    assert one["read_stats.txt"] == two["read_stats.txt"] and len(one["perf.bin"]) == 16 and one["smallContigs/smallContigs_k4.bin"] == b""
    n_rows = int((lens // 271 >= 4).sum())
    assert len(two["kminmerData_abundance.txt"]) == 20 * n_rows > 0
    for name, width in (("kminmerData_abundance.txt", 20), ("kminmerData_abundance_init.txt", 20), ("kminmerData_min.txt", 16)):
        a = np.frombuffer(one[name], np.uint8).reshape(-1, width)
        b = np.frombuffer(two[name], np.uint8).reshape(-1, width)
        assert a.shape == b.shape and np.array_equal(a[np.lexsort(a.T[::-1])], b[np.lexsort(b.T[::-1])]), name
    assert one["kminmerData_abundance.txt"] == one["kminmerData_abundance_init.txt"]
    solid = [ln for ln in two["log"].splitlines() if "Nb solid kminmers" in ln or "Checksum kminmer abundance" in ln]
    assert len(solid) == 2 and [ln for ln in one["log"].splitlines() if "Nb solid kminmers" in ln or "Checksum kminmer abundance" in ln] == solid
    assert "mdbg_tool readSelection" in one["log"] and "mdbg_tool graph" in one["log"]


def test_asm_step_refuses_what_it_cannot_do(stub_tool, read_set, tmp_path):
    fasta = read_set[0]
    tmp = tmp_path / "x" / "tmp"
    os.makedirs(tmp / "filter")
    formats.Parameters(minimizer_size=15, kminmer_size=4, density=0.005, first_k=4, prev_k=4, hpc=False, data_type=1, correction_density=0.025).save(str(tmp / "parameters.gz"))
    (tmp / "input.txt").write_text(fasta + "\n")
    rs = [str(tmp), str(tmp / "read_data_init.txt"), str(tmp / "input.txt"), "--threads", "4", "--min-read-quality", "0.000000"]
    r = subprocess.run([stub_tool, "asmStep"] + rs, capture_output=True, text=True, timeout=60)          # ONT without --skip-correction: read correction lies in between
    assert r.returncode != 0 and "read correction" in r.stderr
    r = subprocess.run([stub_tool, "asmStep"] + rs + ["--skip-correction", "--gpus", "2"], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "one device" in r.stderr
