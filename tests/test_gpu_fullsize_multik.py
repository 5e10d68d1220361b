"""BASELINE.json configs[2]'s loop k = 4 .. 11 at configs[1]'s size (1 M x 10 kb HiFi reads), beyond what the reference can be run on
inside a test -- (3) below ties every one of these tables to digests of the reference's own, made in the build container --:
(1) shard invariance at every k -- the table of the whole set against the union of the shares of its two halves,
through the library's sharded passes and its exchange on the device (bench.shard_self_check: counts and the four sums of
mdbg_table_checksum, sums[0] being the checksum the reference logs, graph/CreateMdbg.cpp:3321); (2) the oracle on the first 2 000
reads' windows at every k > 4 against the WHOLE set's tables: the abundance of a k-min-mer there is a function of its identity and of
the previous table (getRefinedAbundance graph/CreateMdbg.hpp:3933-4005; IndexKminmerFunctor :1240-1265, :1450-1459), so what the
oracle makes of those reads with the whole previous table must be in the whole table, value for value, and the windows it leaves out
must be absent.  GPU box: python -m pytest tests -m gpu"""
from __future__ import annotations

import os
import sys

import numpy as np
import pytest

from metamdbg_amd import synth

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from metamdbg_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


def test_multik_tables_at_one_million_reads(ctx):
    import bench
    from oracle import pyoracle as orc
    import json
    multik = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hifi_1m", "manifest.json")))["multik"]
    # k = 4 .. 11 is BASELINE.json configs[2]; the fixture goes on to the reference's tables of k = 12 .. 24 (make_golden.py --only-1m --multik
    # --last-k 24): the generic window hash behind k >= 12 -- 16-byte blocks, every tail length of Murmur3 -- against the reference at a million reads
    n, n_sample, last_k = 1_000_000, 2_000, int(multik["last_k"])
    assert last_k >= 11
    spec = synth.hifi_spec(n, seed=42, read_len=10_000, coverage=50.0)
    reads = ctx.reads_synthetic(spec)
    mins = ctx.scan(reads, K=15, density=0.005, hpc=True)
    reads.free()
    corr = ctx.purge_palindromes(mins, 4, 100)
    mins.free()
    # (1) every k, the whole set against its two shards (and against three: another cut, another owner map)
    for shards in (2, 3):
        r = bench.shard_self_check(ctx, corr, list(range(4, last_k + 1)), n_shards=shards)
        assert r["all_equal"], r
        assert r["per_k"]["4"]["records"] == 1_080_243 and r["per_k"]["4"]["solid"] == 1_064_017       # tests/golden/hifi_1m: the reference's own counts
    # (2) the oracle on the first reads against the whole set's tables, and (3) every table whole against THE REFERENCE: the digests of what
    # its own `graph` wrote at k = 4 .. 11 on this read set in the same mode (tests/golden/hifi_1m/manifest.json "multik", made by
    # tests/golden/make_golden.py --only-1m --multik: the previous table is the reference's own table of k - 1, no unitigs) -- record count,
    # sha256 of the sorted 20-byte records (of the sorted vectors at k <= 5), the checksum the reference logs, the sum of abundances
    from metamdbg_amd import formats
    golden = multik["per_k"]

    def assert_reference_digests(table, k):
        rec, vec = table.to_host()
        g = golden[str(k)]
        assert len(rec) == g["n_records"] and int(rec["abundance"].astype(np.uint64).sum()) == g["sum_abundance"], k
        assert table.checksum()[0] == g["abundance_checksum"], k
        mine = formats.table_digests(rec, vec.astype("<u4").tobytes() if "min_sorted_sha256" in g else None, k)
        assert mine == {key: g[key] for key in mine}, k

    head = ctx.minimizers_slice(corr, 0, n_sample).to_host(full=False)
    m, off = head["minimizers"], head["offsets"].astype(np.int64)
    prev = ctx.kminmer_count_first(corr, 4, 0)
    assert_reference_digests(prev, 4)
    for k in range(5, last_k + 1):
        whole = ctx.kminmer_count_refined(corr, None, k, prev) if k == 5 else ctx.kminmer_index(corr, None, k, prev)
        assert_reference_digests(whole, k)
        pa = orc.PrevAbundance(prev.to_host()[0].tobytes())
        exp = (orc.kminmer_count_refined if k == 5 else orc.kminmer_index)(m, head["offsets"], k, pa)
        assert exp["n"] > (1000 if k <= 11 else 200), (k, exp["n"])
        got = whole.lookup(exp["hash_lo"], exp["hash_hi"])
        assert np.array_equal(got, exp["abundance"]), k
        # the windows of these reads the oracle did not list are not in the whole table either
        listed = set(zip(exp["hash_lo"].tolist(), exp["hash_hi"].tolist()))
        lo, hi = [], []
        for r0 in range(n_sample):
            a, b = int(off[r0]), int(off[r0 + 1])
            for i in range(a, b - k + 1):
                _, _, h_hi, h_lo = orc.kminmer_normalize_hash(m[i: i + k])
                if (h_lo, h_hi) not in listed:
                    lo.append(h_lo); hi.append(h_hi)
        if lo:
            assert not whole.lookup(np.array(lo, np.uint64), np.array(hi, np.uint64)).any(), k
        prev.free()
        prev = whole
    prev.free()
    corr.free()


def test_configs2_at_its_stated_size_ten_million_reads(ctx):
    """BASELINE.json configs[2] at the size it names -- 10 M x 10 kb HiFi reads, k = 4 .. 11 -- inside the GPU suite (until round 4 only
    bench.py ran it): at every k the table of the whole read set against the union of the shares of its halves, through the library's
    sharded passes and its exchange (bench.shard_self_check).  The k = 4 counts are the ones every bench line of rounds 3 - 5 reports."""
    import bench
    ctx.set_option("pool_trim", 1)
    spec = synth.hifi_spec(10_000_000, seed=42, read_len=10_000, coverage=50.0)
    reads = ctx.reads_synthetic(spec)
    mins = ctx.scan(reads, K=15, density=0.005, hpc=True)
    reads.free()
    assert mins.info()["n_minimizers"] == 373_753_601
    corr = ctx.purge_palindromes(mins, 4, 100)
    mins.free()
    r = bench.shard_self_check(ctx, corr, list(range(4, 12)), n_shards=2)
    corr.free()
    ctx.set_option("pool_trim", 1)
    assert r["all_equal"] and r["reads"] == 10_000_000, r
    assert r["per_k"]["4"]["records"] == 10_775_506 and r["per_k"]["4"]["solid"] == 10_613_396
    assert all(r["per_k"][str(k)]["records"] > 11_000_000 for k in range(5, 12))


def test_configs3_at_its_stated_size_ten_million_ont_reads(ctx):
    """BASELINE.json configs[3] at the size it names -- 10 M x 20 kb ONT reads with qualities (200 Gbp, resident three pieces at a
    time), nanoMDBG parameters -- inside the GPU suite: bench.ont_leg without its reference sample (tests/test_gpu_parity.py and the
    bench line compare samples with the reference): census, scans, purge and the first pass over all the reads, and the 752 M-row
    table against the union of the shares of its halves (ont_leg raises when they differ)."""
    import bench
    from metamdbg_amd import capi
    ctx.set_option("pool_trim", 1)
    info = ctx.device_info()
    if info["hbm_bytes"] < 250e9:
        pytest.skip("needs an MI355X's 288 GB")
    try:                                   # what is free right now (hipMemGetInfo: the runtime the library itself is linked against)
        import ctypes as C
        hip = C.CDLL("libamdhip64.so")
        free, total = C.c_size_t(), C.c_size_t()
        if hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0 and free.value < 240e9:
            pytest.skip(f"only {free.value / 1e9:.0f} GB of the device are free")
    except OSError:
        pass
    octx = capi.Context(0)
    try:
        r = bench.ont_leg(octx, 10_000_000, sample=0)
    finally:
        octx.close()
    assert r["bases"] == 200_000_000_000 and r["self_check"]["all_equal"] and r["self_check"]["reads"] == 10_000_000
    assert r["kminmer_records"] > 700_000_000 and r["first_pass"]["path"] == 2 and r["pieces"] == 3
