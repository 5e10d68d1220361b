"""The partitioned first pass (csrc/partition.hip: instances split by key, every bucket counted in LDS, keys in groups) against the
oracle and against the one-table path, through the C ABI.  The reference's analogue is KminmerCounter's partition -> sort ->
run-length (graph/CreateMdbg.hpp:3714-3851, _nbPartitions from graph/CreateMdbg.cpp:222-225).  GPU box: python -m pytest tests -m gpu"""
from __future__ import annotations

import numpy as np
import pytest

from metamdbg_amd import formats, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from metamdbg_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def orc():
    from oracle import pyoracle
    return pyoracle


def _options(ctx, **kw):
    for name in ("first_pass_mode", "partition_bits", "partition_lds_slots", "partition_max_records"):
        ctx.set_option(name, kw.get(name, 0))


def _random_minimizer_reads(rng, n_reads, alphabet, lo=0, hi=60):
    lens = rng.integers(lo, hi, n_reads)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    mins = rng.integers(0, alphabet, int(offs[-1])).astype(np.uint32)
    return mins, offs


# (bucket bits, LDS slots, instances per group): one level, two, three; a table of 256 slots that must overflow and repeat; groups
PLANS = [dict(), dict(partition_bits=3), dict(partition_bits=10), dict(partition_bits=17, partition_lds_slots=256),
         dict(partition_bits=3, partition_lds_slots=256), dict(partition_max_records=1500), dict(partition_max_records=300, partition_bits=9)]


@pytest.mark.parametrize("plan", range(len(PLANS)))
@pytest.mark.parametrize("k", [2, 3, 4, 5, 8, 12])
@pytest.mark.parametrize("min_ab", [0, 2, 3])
def test_partitioned_count_first_vs_oracle(ctx, orc, k, min_ab, plan):
    """Small alphabets force palindromic windows, repeated keys and the rescue pass's tie case; empty and short reads are in."""
    rng = np.random.default_rng(1000 + 17 * k + plan)
    mins, offs = _random_minimizer_reads(rng, 400, 6 if k >= 8 else 25)
    _options(ctx, first_pass_mode=2, **PLANS[plan])
    try:
        t = ctx.kminmer_count_first(ctx.minimizers_from_host(mins, offs), k, min_ab)
        info = ctx.first_pass_info()
    finally:
        _options(ctx)
    rec, vec = t.to_host()
    exp = orc.kminmer_count_first(mins, offs, k, min_ab)
    assert info["path"] == 2, info
    n_inst = int(np.maximum(np.diff(offs.astype(np.int64)) - (k - 1), 0).sum())
    assert info["instances"] == n_inst and t.stats()["instances"] == n_inst
    if "partition_max_records" in PLANS[plan]:
        assert info["groups"] > 1, info
    if PLANS[plan].get("partition_lds_slots") == 256 and PLANS[plan].get("partition_bits") == 3 and k >= 3:
        assert info["attempts"] > 1, info                    # 8 x 256 slots cannot hold the keys of 10^4 windows over 25 values
    assert t.info()["n_solid"] == exp["n_solid"]
    assert np.array_equal(formats.sorted_abundance_records(rec), formats.sorted_abundance_records(orc.table_abundance_records(exp)))
    assert np.array_equal(formats.sorted_vector_records(vec.astype("<u4").tobytes(), k),
                          formats.sorted_vector_records(exp["vecs"].astype("<u4").tobytes(), k))
    # file order: solid rows, then the rescued ones in read order
    assert (rec[: exp["n_solid"]]["abundance"] > 1).all() and (rec[exp["n_solid"]:]["abundance"] == 1).all()
    n_res = len(rec) - exp["n_solid"]
    if n_res:
        assert np.array_equal(vec[exp["n_solid"]:], exp["vecs"][exp["n_solid"]:])


def test_partitioned_rescue_tie_case(ctx, orc):
    """Reads whose even number of windows splits exactly in half around m* = 10: the decision needs exact counts up to 2 m* + 1
    (Utils::compute_median on u32, Commons.hpp:2972-2988; `median * 0.1f > 1`, graph/CreateMdbg.hpp:4610).  A probe read of six
    distinct minimizers has four 3-windows with counts {1, a, b, c}, a <= 10 < b, c: it is rescued -- its count-1 window appended --
    iff (a + min(b, c)) / 2 <= 10.  The counts are made by single-window reads."""
    k = 3
    rows, nxt, expect = [], 1000, 0
    for a in (2, 5, 9, 10):
        for b, c in ((11, 11), (12, 30), (11, 22), (13, 21), (15, 300), (21, 22), (22, 23), (40, 11), (10 + a, 25), (21 - a, 21 - a), (22 - a, 22 - a)):
            if b <= 10 or c <= 10:
                continue
            x = list(range(nxt, nxt + 6)); nxt += 10
            rows.append(x)
            for w, cnt in ((1, a), (2, b), (3, c)):
                rows.extend([x[w: w + 3]] * (cnt - 1))
            expect += (a + min(b, c)) // 2 <= 10
    rng = np.random.default_rng(3)
    rows = [rows[i] for i in rng.permutation(len(rows))]
    mins = np.array([v for r in rows for v in r], dtype=np.uint32)
    offs = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.uint64)
    exp = orc.kminmer_count_first(mins, offs, k, 0)
    assert len(exp["vecs"]) - exp["n_solid"] == expect and 0 < expect < 40
    for plan in (dict(), dict(partition_bits=6), dict(partition_max_records=2000)):
        _options(ctx, first_pass_mode=2, **plan)
        try:
            t = ctx.kminmer_count_first(ctx.minimizers_from_host(mins, offs), k, 0)
            assert ctx.first_pass_info()["path"] == 2
        finally:
            _options(ctx)
        rec, vec = t.to_host()
        assert t.info()["n_solid"] == exp["n_solid"]
        assert np.array_equal(formats.sorted_abundance_records(rec), formats.sorted_abundance_records(orc.table_abundance_records(exp)))
        assert np.array_equal(vec[exp["n_solid"]:], exp["vecs"][exp["n_solid"]:])


@pytest.mark.parametrize("n_reads,plan", [(200_000, dict()), (200_000, dict(partition_max_records=1_000_000)), (60_000, dict(partition_bits=20)),
                                          (1_000_000, dict())])
def test_partitioned_equals_one_table_on_hifi_reads(ctx, n_reads, plan):
    """Synthetic HiFi reads (50 x): the partitioned pass and the one-table pass give the same table -- rows as multisets, the
    rescued rows in the same (read) order, the four order-independent sums, and the pass's statistics."""
    spec = synth.hifi_spec(n_reads, seed=7, read_len=10_000, coverage=50.0)
    reads = ctx.reads_synthetic(spec)
    corr = ctx.purge_palindromes(ctx.scan(reads, K=15, density=0.005, hpc=True), 4, 100)
    reads.free()
    out = {}
    for mode in (1, 2):
        _options(ctx, first_pass_mode=mode, **(plan if mode == 2 else {}))
        try:
            t = ctx.kminmer_count_first(corr, 4, 0)
            info = ctx.first_pass_info()
        finally:
            _options(ctx)
        assert info["path"] == mode, info
        if mode == 2 and "partition_max_records" in plan:
            assert info["groups"] > 1
        rec, vec = t.to_host()
        out[mode] = (rec, vec, t.info(), t.stats(), t.checksum())
        t.free()
    a, b = out[1], out[2]
    assert a[2] == b[2]
    assert a[3]["instances"] == b[3]["instances"] and a[3]["keys"] == b[3]["keys"] and a[3]["minimizers"] == b[3]["minimizers"]
    assert list(a[4]) == list(b[4])
    ns = a[2]["n_solid"]
    assert np.array_equal(formats.sorted_abundance_records(a[0]), formats.sorted_abundance_records(b[0]))
    assert np.array_equal(formats.sorted_vector_records(a[1][:ns].astype("<u4").tobytes(), 4),
                          formats.sorted_vector_records(b[1][:ns].astype("<u4").tobytes(), 4))
    assert np.array_equal(a[1][ns:], b[1][ns:]) and np.array_equal(a[0][ns:], b[0][ns:])


def test_partitioned_hands_back_what_it_does_not_take(ctx, orc):
    """k > 32 and inputs shorter than a window go to the one-table path whatever the mode says."""
    rng = np.random.default_rng(5)
    mins, offs = _random_minimizer_reads(rng, 50, 4, lo=30, hi=80)
    _options(ctx, first_pass_mode=2)
    try:
        t = ctx.kminmer_count_first(ctx.minimizers_from_host(mins, offs), 33, 0)
        assert ctx.first_pass_info()["path"] == 1
        exp = orc.kminmer_count_first(mins, offs, 33, 0)
        rec, vec = t.to_host()
        assert np.array_equal(formats.sorted_abundance_records(rec), formats.sorted_abundance_records(orc.table_abundance_records(exp)))
        t2 = ctx.kminmer_count_first(ctx.minimizers_from_host(mins[:3], np.array([0, 1, 3], dtype=np.uint64)), 4, 0)
        assert t2.info()["n_records"] == 0
    finally:
        _options(ctx)


@pytest.mark.parametrize("n_ranks", [1, 2, 3, 8])
@pytest.mark.parametrize("plan", [0, 2, 4, 5])
@pytest.mark.parametrize("k,min_ab", [(3, 0), (4, 0), (4, 2), (8, 0)])
def test_partitioned_share_of_a_sharded_first_pass_vs_oracle(ctx, orc, n_ranks, plan, k, min_ab):
    """A rank's share counted by the partitioned pass (mdbg_shard_begin takes it like mdbg_kminmer_count_first does): its distinct keys go to
    their owners as rows, the global counts come back, the records are walked a second time for the rescue pass against the global counts.
    Ranks alternate between the partitioned and the one-table local pass; the union of the shares must be the oracle's table of all reads,
    the rescued rows of every rank in its own read order."""
    from metamdbg_amd import capi
    rng = np.random.default_rng(7000 + 31 * k + plan + n_ranks)
    mins, offs = _random_minimizer_reads(rng, 600, 6 if k >= 8 else 25)
    exp = orc.kminmer_count_first(mins, offs, k, min_ab)
    cuts = [600 * r // n_ranks for r in range(n_ranks + 1)]
    whole = ctx.minimizers_from_host(mins, offs)
    parts = [ctx.minimizers_slice(whole, cuts[r], cuts[r + 1] - cuts[r]) for r in range(n_ranks)]
    shards = []
    try:
        for r, part in enumerate(parts):
            _options(ctx, first_pass_mode=1 if (r % 2 == 1 and n_ranks > 1) else 2, **PLANS[plan])
            shards.append(ctx.shard_begin(part, k, n_ranks))
            if r % 2 == 0 or n_ranks == 1:
                assert ctx.first_pass_info()["path"] == 2
        replies = capi.exchange_local(ctx, shards)
        tables = [sh.finish(rep, min_ab) for sh, rep in zip(shards, replies)]
    finally:
        _options(ctx)
    recs, vecs, n_solid = [], [], 0
    for t in tables:
        rec, vec = t.to_host()
        ns = t.info()["n_solid"]
        assert (rec[:ns]["abundance"] > 1).all() and (rec[ns:]["abundance"] == 1).all()
        recs.append(rec); vecs.append(vec); n_solid += ns
    assert n_solid == exp["n_solid"]
    assert np.array_equal(formats.sorted_abundance_records(np.concatenate(recs)), formats.sorted_abundance_records(orc.table_abundance_records(exp)))
    assert np.array_equal(formats.sorted_vector_records(np.concatenate(vecs).astype("<u4").tobytes(), k),
                          formats.sorted_vector_records(exp["vecs"].astype("<u4").tobytes(), k))
    # the rescued rows: every rank's in the order of its own reads, the ranks' read ranges in order = the oracle's read order
    resc = np.concatenate([v[t.info()["n_solid"]:] for v, t in zip(vecs, tables)])
    assert np.array_equal(resc, exp["vecs"][exp["n_solid"]:])
