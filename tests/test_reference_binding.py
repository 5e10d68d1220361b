"""oracle/_ref/refdrv_hip: the binding of INTEGRATION.md compiled INTO the reference's own tools (oracle/ref_binding.cpp -- classes derived
from the reference's ReadSelection / CreateMdbg whose compute goes through include/mdbg_hip.h; everything around it, parser, ordered
writer, statistics, graph stage, is the reference's code).

CPU part (here): the plumbing, against the TEST DOUBLE of the library (tests/host/stub_mdbg_hip.cpp, fake minimizers that depend on a
read's length only) -- the reference's ReadParserParallel feeding batches through the C ABI and the reference's writeRead putting the
records in read order must give exactly the bytes the fake minimizers imply, for any batch size and thread count.
GPU part: tests/test_gpu_reference_binding.py (the real library, compared with the unmodified reference on the same inputs)."""
from __future__ import annotations

import os
import struct
import subprocess

import numpy as np
import pytest

from metamdbg_amd import formats

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDRV_HIP = os.path.join(ROOT, "oracle", "_ref", "refdrv_hip")

pytestmark = pytest.mark.skipif(not os.path.exists(REFDRV_HIP), reason="oracle/_ref/refdrv_hip not built (needs /root/reference and the library)")


@pytest.fixture(scope="module")
def stub_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("stublib")
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", os.path.join(ROOT, "tests", "host", "stub_mdbg_hip.cpp"), "-o",
                    str(d / "libmdbg_hip.so"), "-lpthread"], check=True)
    return str(d)


@pytest.mark.parametrize("batch_bases,threads,fastq", [(1 << 16, 4, False), (1 << 20, 8, False), (1 << 26, 3, False), (1 << 18, 6, True)])
def test_reference_tool_with_the_binding_orders_and_completes(stub_dir, tmp_path, batch_bases, threads, fastq):
    rng = np.random.default_rng(11)
    lens = rng.integers(1, 6000, 4000).astype(np.int64)
    pool = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 100_000)]
    path = str(tmp_path / ("reads.fastq" if fastq else "reads.fasta"))
    with open(path, "wb") as f:
        for i, n in enumerate(lens):
            a = int(rng.integers(0, len(pool) - 6000))
            s = pool[a:a + int(n)].tobytes()
            f.write((b"@r%d\n%s\n+\n%s\n" % (i, s, b"I" * int(n))) if fastq else (b">r%d\n%s\n" % (i, s)))
    init, corr = [], []
    for n in lens:
        L, k = int(n), int(n) // 271
        m = ((L * 2654435761 + np.arange(k, dtype=np.uint64) * 40503) & 0xFFFFFFFF).astype("<u4")
        init.append(struct.pack("<IB", k, 0) + m.tobytes() + (np.arange(k, dtype="<u4") * 271).astype("<u4").tobytes() +
                    ((L + np.arange(k)) & 1).astype("u1").tobytes() + np.ones(k, "u1").tobytes() + struct.pack("<II", 0xFFC00000, L))
        corr.append(struct.pack("<IB", k, 0) + m.tobytes())
    tmp = tmp_path / "asm" / "tmp"
    os.makedirs(tmp / "filter")
    formats.Parameters(minimizer_size=15, kminmer_size=4, density=0.005, first_k=4, prev_k=4, hpc=True, data_type=0).save(str(tmp / "parameters.gz"))
    (tmp / "input.txt").write_text(path + "\n")
    env = dict(os.environ, LD_LIBRARY_PATH=stub_dir, MDBG_BINDING_BATCH_BASES=str(batch_bases))
    r = subprocess.run([REFDRV_HIP, "readSelection_hip", str(tmp), str(tmp / "read_data_init.txt"), str(tmp / "input.txt"), "--threads", str(threads),
                        "--min-read-quality", "0.0"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-800:]
    assert (tmp / "read_data_init.txt").read_bytes() == b"".join(init)
    # the reference's own purge pass writes in completion order; the binding's, batch by batch in arrival order: compare as multisets
    got = (tmp / "read_data_corrected.txt").read_bytes()
    assert len(got) == sum(map(len, corr))
    recs, o = [], 0
    while o < len(got):
        k = struct.unpack_from("<I", got, o)[0]
        recs.append(got[o:o + 5 + 4 * k])
        o += 5 + 4 * k
    assert sorted(recs) == sorted(corr)
    st = formats.parse_read_stats((tmp / "read_stats.txt").read_bytes())
    assert st["n_reads"] == len(lens) and st["n_bases"] == int(lens.sum()) and st["n_minimizers"] == int((lens // 271).sum())
    assert os.path.getsize(tmp / "perf.bin") == 16            # Tool::end ran (Commons.hpp:8088-8107)
