"""HIP path vs the CPU oracle and the reference's golden vectors, through the C ABI.
Run on the GPU box: python -m pytest tests -m gpu"""
from __future__ import annotations

import json
import os

import numpy as np
import pytest

from metamdbg_amd import formats, synth
from tests import helpers as H

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


def _nan_eq(a, b):
    a = np.asarray(a, np.float32).view(np.uint32)
    b = np.asarray(b, np.float32).view(np.uint32)
    return np.array_equal(a, b)


def test_device_info(ctx):
    info = ctx.device_info()
    assert info["arch"].startswith("gfx950") and info["n_cu"] >= 200


def test_synthetic_generator_matches_numpy(ctx):
    spec = synth.hifi_spec(64, seed=5, read_len=3000, coverage=10.0)
    reads = ctx.reads_synthetic(spec)
    codes = synth.read_codes(spec, 0, spec.n_reads)
    asc = synth.codes_to_ascii(codes)
    for r in (0, 1, 17, 63):
        assert reads.get(r) == asc[r].tobytes()
    spec.with_quality = True
    reads = ctx.reads_synthetic(spec, first_read=10, n_reads=5)
    q = synth.read_qualities(spec, 10, 15)
    b, qq = reads.get(3, with_quality=True)
    assert b == asc[13].tobytes() and qq == q[3].tobytes()


def test_synthetic_generator_past_2_32_words(ctx):
    """14 M reads x 313 words are more than 2^32 words: a launch of one thread per word wraps around at 2^32 threads and leaves the reads
    behind it zero (found as 11.8 instead of 37.4 minimizers per read in a 40 M-read pass).  The last read, and one just past the wrap."""
    n = 14_000_000
    spec = synth.hifi_spec(n, seed=5, read_len=10_000, coverage=50_000.0)
    reads = ctx.reads_synthetic(spec)
    first_wrapped = (1 << 32) // 313 + 1
    for r in (0, first_wrapped, n - 1):
        assert reads.get(r) == synth.codes_to_ascii(synth.read_codes(spec, r, r + 1))[0].tobytes(), r
    del reads


def _check_scan_against_oracle(ctx, orc, seqs, quals, K, density, hpc, repetitive=None):
    reads = ctx.reads_from_ascii(seqs, quals)
    m = ctx.scan(reads, K=K, density=density, hpc=hpc, repetitive=repetitive)
    h = m.to_host()
    exp = b"".join(orc.read_selection(s, quals[i] if quals else None, K=K, density=density, hpc=hpc,
                                      repetitive=repetitive)["record"] for i, s in enumerate(seqs))
    got = formats.build_read_data_init(h)
    assert got == exp
    return m, h


@pytest.mark.parametrize("read_len", [20_000, 777, 65_536])
def test_synthetic_generator_with_indels_matches_numpy(ctx, read_len):
    """ONT R10 error model (1 % substitutions + 0.5 % insertions + 0.5 % deletions): the device generator's
    per-word event counts and scan must land every read position on the genome base the numpy twin picks."""
    spec = synth.SynthSpec(n_reads=70, read_len=read_len, seed=29, sub_rate=0.01, ins_rate=0.005, del_rate=0.005,
                           species_len=[3 * read_len + 5000, 2 * read_len + 900], species_weight=[0.6, 0.4], with_quality=True, name="ont")
    reads = ctx.reads_synthetic(spec, first_read=3, n_reads=60)
    asc = synth.codes_to_ascii(synth.read_codes(spec, 3, 63))
    q = synth.read_qualities(spec, 3, 63)
    for r in (0, 1, 2, 31, 59):
        b, qq = reads.get(r, with_quality=True)
        assert b == asc[r].tobytes() and qq == q[r].tobytes(), r
    # the model really shifts the reads against the genome
    clean = synth.SynthSpec(**{**spec.__dict__, "sub_rate": 0.0, "ins_rate": 0.0, "del_rate": 0.0})
    assert spec.window() > read_len and clean.window() == read_len
    if read_len == 20_000:      # error rates: about 1 % + 0.5 % + 0.5 % of the positions draw an event
        g = synth.genome_codes(spec)
        e_ins = e_del = 0
        codes = synth.read_codes(spec, 0, 20, g)
        import numpy as _np
        r = _np.arange(0, 20, dtype=_np.uint64)[:, None]
        with _np.errstate(over="ignore"):
            rk = synth.mix64(_np.uint64(spec.seed) ^ _np.uint64(0x5EED5EED5EED5EED)) + r
            e = synth.mix64(synth.mix64(rk) + _np.arange(read_len, dtype=_np.uint64)[None, :])
        f_ins = float((e < _np.uint64(spec.ins_threshold())).mean())
        f_del = float(((e >= _np.uint64(spec.ins_threshold())) & (e < _np.uint64(spec.ins_threshold() + spec.del_threshold()))).mean())
        assert 0.004 < f_ins < 0.006 and 0.004 < f_del < 0.006 and codes.shape == (20, read_len)


def test_repetitive_minimizers_content(ctx, orc):
    """SURVEY 8(a) A8, determineRepetitiveMinimizers (readSelection/ReadSelection.hpp:497-625): the device census' CONTENT.
    (1) ont_rep: a read set for which the reference's answer is unambiguous (several minimizers to pick, no count tied
    across the cut): the device must return exactly the reference's set.  (2) ont_100: the census recomputed with the oracle;
    every value returned must have a count at or above the cut-off count, everything above it must be returned, and the set
    has max(1, floor(1e-5 f x distinct)) members (ties AT the cut are the reference's std::sort order, unspecified)."""
    import ctypes as C
    m = H.load_manifest("ont_rep")
    spec = H.spec_from_manifest(m)
    reads = ctx.reads_synthetic(spec)
    pre = ctx.scan(reads, K=m["K"], density=m["correction_density"], hpc=False, apply_read_filters=False)
    rep_gpu = ctx.repetitive_minimizers(pre)
    rep_ref = np.frombuffer(H.golden_bytes("ont_rep", "repetitiveMinimizers.bin"), "<u4")
    assert len(rep_gpu) == m["n_keep"] >= 3 and len(set(rep_gpu.tolist())) == len(rep_gpu)
    assert set(rep_gpu.tolist()) == set(rep_ref.tolist())
    # the counts behind it, from the scan's own minimizer list (which other tests pin against the oracle)
    vals, counts = np.unique(pre.to_host(full=False)["minimizers"], return_counts=True)
    assert len(vals) == m["n_distinct"]
    got = dict(zip(vals.tolist(), counts.tolist()))
    assert min(got[int(v)] for v in rep_gpu) == m["cut_count"] and m["cut_count"] > m["next_count"]
    pre.free(); reads.free()
    # (2) oracle census, ties allowed
    m = H.load_manifest("ont_100")
    seqs, _ = H.regenerate_reads(m)
    L = orc.lib()
    L.orc_minimizer_parse.restype = C.c_size_t
    allm = []
    for sq in seqs:
        n = len(sq)
        om = (C.c_uint32 * n)(); op = (C.c_uint32 * n)(); od = (C.c_uint8 * n)()
        k = L.orc_minimizer_parse(sq, C.c_size_t(n), 15, C.c_float(0.025), None, C.c_size_t(0), om, op, od)
        allm.append(np.frombuffer(om, np.uint32, k).copy())
    vals, counts = np.unique(np.concatenate(allm), return_counts=True)
    n_keep = max(int(np.float32(0.00001) * np.float32(len(vals))), 1)
    cut = np.sort(counts)[::-1][n_keep - 1]
    reads = ctx.reads_from_ascii(seqs)
    rep_gpu = ctx.repetitive_minimizers(ctx.scan(reads, K=15, density=0.025, hpc=False, apply_read_filters=False))
    cnt = dict(zip(vals.tolist(), counts.tolist()))
    assert len(rep_gpu) == n_keep and len(set(rep_gpu.tolist())) == n_keep
    assert all(cnt[int(v)] >= cut for v in rep_gpu)
    assert set(vals[counts > cut].tolist()) <= set(rep_gpu.tolist())


@pytest.mark.parametrize("hpc", [True, False])
@pytest.mark.parametrize("K,density", [(15, 0.005), (16, 0.02), (13, 0.05), (11, 0.01)])
def test_scan_random_reads_vs_oracle(ctx, orc, hpc, K, density):
    rng = np.random.default_rng(K * 7 + int(hpc))
    seqs = [bytes(synth.CODE2ASCII[rng.integers(0, 4, int(n))]) for n in
            list(rng.integers(1, 200, 40)) + list(rng.integers(200, 9000, 40)) + [2048, 2049, 2047, 4096, 4111, 64, 65, 66, 67]]
    _check_scan_against_oracle(ctx, orc, seqs, None, K, density, hpc)


@pytest.mark.parametrize("hpc", [True, False])
@pytest.mark.parametrize("K,density", [(15, 1.0), (15, 0.99999994), (13, 1.0), (16, 0.75)])
def test_scan_densities_at_and_near_one_vs_oracle(ctx, orc, hpc, K, density):
    """Densities whose threshold lies within a few 2^32 of 2^64 (1.0f: T = 2^64 - 1024, so hi(T) + 3 does not fit 32 bits) take the block
    kernel's FULL verdict -- the candidate limit of span_step<APPROX> would saturate and a hash with upper half 0xFFFFFFFF slip through --,
    both half spans of a lane at once (scan.hip, aligned_block); just below (0.99999994f, the float before 1) and at 0.75 the candidate
    test runs with nearly every position a candidate, more per block than the stage holds.  Reads long enough for several blocks and a tail."""
    rng = np.random.default_rng(K * 11 + int(hpc) + int(density * 8))
    seqs = [bytes(synth.CODE2ASCII[rng.integers(0, 4, int(n))]) for n in
            list(rng.integers(1, 200, 10)) + list(rng.integers(2100, 9000, 10)) + [2048, 2049, 4096 + K, 4096 + K + 1, 6200]]
    _check_scan_against_oracle(ctx, orc, seqs, None, K, density, hpc)


@pytest.mark.parametrize("tag,K,dens,hpc", [("hpc_k15", 15, 0.005, True), ("nohpc_k15", 15, 0.005, False),
                                            ("hpc_k16", 16, 0.005, True), ("nohpc_k13", 13, 0.02, False)])
def test_scan_edge_reads_golden(ctx, tag, K, dens, hpc):
    seqs = [s.upper() for s in H.read_fasta(os.path.join(H.GOLDEN, "edge", "edge.fasta"))]
    # lower-case read: the reference's HPC compares raw chars; upper-casing is equivalent here because
    # the read is uniformly lower-case (see DESIGN.md "character handling")
    rep = np.frombuffer(H.golden_bytes("edge", f"repetitiveMinimizers.{tag}.bin"), "<u4")
    reads = ctx.reads_from_ascii(seqs)
    h = ctx.scan(reads, K=K, density=dens, hpc=hpc, repetitive=rep).to_host()
    assert formats.build_read_data_init(h) == H.golden_bytes("edge", f"read_data_init.{tag}.txt")


def test_hifi_200_golden_end_to_end(ctx, orc):
    m = H.load_manifest("hifi_200")
    spec = H.spec_from_manifest(m)
    reads = ctx.reads_synthetic(spec)               # generated in HBM, same reads as the fixture's FASTA
    seqs, _ = H.regenerate_reads(m)
    assert reads.get(7) == seqs[7] and reads.get(199) == seqs[199]
    mins = ctx.scan(reads, K=m["K"], density=m["density"], hpc=m["hpc"])
    h = mins.to_host()
    assert formats.build_read_data_init(h) == H.golden_bytes("hifi_200", "read_data_init.txt")
    st = formats.parse_read_stats(H.golden_bytes("hifi_200", "read_stats.txt"))
    last_k = orc.lib().orc_compute_last_k(m["density"], st["n50"], 4, 0)
    corr = ctx.purge_palindromes(mins, 4, last_k)
    hc = corr.to_host(full=False)
    assert formats.write_minimizer_reads(hc["minimizers"], hc["offsets"]) == H.golden_bytes("hifi_200", "read_data_corrected.txt")
    t = ctx.kminmer_count_first(corr, m["k"], m["min_abundance"])
    rec, vec = t.to_host()
    exp_ab = np.fromfile(os.path.join(H.GOLDEN, "hifi_200", "kminmerData_abundance.sorted.bin"), formats.ABUNDANCE_DTYPE)
    assert np.array_equal(formats.sorted_abundance_records(rec), exp_ab)
    exp_v = np.fromfile(os.path.join(H.GOLDEN, "hifi_200", "kminmerData_min.sorted.bin"), "<u4").reshape(-1, m["k"])
    assert np.array_equal(formats.sorted_vector_records(vec.astype("<u4").tobytes(), m["k"]), exp_v)


def test_purge_palindromes_fn_golden(ctx):
    with open(os.path.join(H.GOLDEN, "fn", "fn_golden.json")) as f:
        g = json.load(f)["purge"]
    for key, gg in g.items():
        fk, lk = map(int, key.split("_"))
        lists = [[int(x) for x in line.split()] for line in gg["inputs"]]
        offs = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.uint64)
        mins = np.array([x for lst in lists for x in lst], dtype=np.uint32)
        out = ctx.purge_palindromes(ctx.minimizers_from_host(mins, offs), fk, lk).to_host(full=False)
        for i, line in enumerate(gg["outputs"]):
            got = out["minimizers"][int(out["offsets"][i]): int(out["offsets"][i + 1])].tolist()
            assert got == [int(x) for x in line.split()], (key, i)


def _random_minimizer_reads(rng, n_reads, alphabet, lo=0, hi=60):
    lens = rng.integers(lo, hi, n_reads)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    mins = rng.integers(0, alphabet, int(offs[-1])).astype(np.uint32)
    return mins, offs


def _assert_tables_equal(rec, vec, t, k):
    from oracle import pyoracle as orc
    assert np.array_equal(formats.sorted_abundance_records(rec), formats.sorted_abundance_records(orc.table_abundance_records(t)))
    if t["vecs"] is not None:
        assert np.array_equal(formats.sorted_vector_records(vec.astype("<u4").tobytes(), k),
                              formats.sorted_vector_records(t["vecs"].astype("<u4").tobytes(), k))


@pytest.mark.parametrize("k", [2, 3, 4, 5, 7, 8, 12])
@pytest.mark.parametrize("min_ab", [0, 2, 3])
def test_count_first_vs_oracle(ctx, orc, k, min_ab):
    rng = np.random.default_rng(100 + k)
    mins, offs = _random_minimizer_reads(rng, 400, 6 if k >= 7 else 25)
    t = ctx.kminmer_count_first(ctx.minimizers_from_host(mins, offs), k, min_ab)
    rec, vec = t.to_host()
    exp = orc.kminmer_count_first(mins, offs, k, min_ab)
    assert t.info()["n_solid"] == exp["n_solid"]
    _assert_tables_equal(rec, vec, exp, k)


def _genome_reads(rng, genome_len, n_reads, max_len, n_err=40, genome=None):
    """Minimizer-space reads = substrings of a "genome" of distinct minimizers in either orientation, a quarter of them with one
    "error"; lengths 0 .. max_len, so reads shorter than k are common."""
    if genome is None:
        genome = rng.permutation(genome_len).astype(np.uint32)
    rl = []
    for _ in range(n_reads):
        n = int(rng.integers(0, max_len)); a = int(rng.integers(0, max(1, genome_len - n)))
        seg = genome[a:a + n]
        if rng.integers(0, 2): seg = seg[::-1]
        seg = seg.copy()
        if len(seg) and rng.integers(0, 4) == 0: seg[int(rng.integers(0, len(seg)))] = genome_len + 1000 + int(rng.integers(0, n_err))
        rl.append(seg)
    offs = np.concatenate([[0], np.cumsum([len(x) for x in rl])]).astype(np.uint64)
    return genome, np.concatenate(rl).astype(np.uint32), offs


@pytest.mark.parametrize("k", [13, 14, 15, 16, 31, 32, 33, 64, 65, 100, 200])
@pytest.mark.parametrize("mode", ["auto", "one_table", "partitioned"])
def test_count_first_deep_k_vs_oracle(ctx, orc, k, mode):
    """The counting pass at the k the reference's default loop reaches (k = 4 .. N50 x density x 2: Commons.hpp:1726-1741): the generic
    window hash (csrc/kminmer_dev.hpp: Murmur tails of 1, 2, 3, 0 words at k = 13, 14, 15, 16), k at and above the partitioned pass's
    limit of 32 (it hands k > 32 to the one-table pass), reads with fewer than k minimizers as the common case, palindromic windows."""
    rng = np.random.default_rng(900 + k)
    _, mins, offs = _genome_reads(rng, 3000, 500, int(2.2 * k) + 20)
    # palindromic and tie-at-first-compare windows (KmerVec::normalize, Commons.hpp:886-916: equal => reversed)
    half = rng.integers(0, 3000, (k + 1) // 2).astype(np.uint32)
    pal = np.concatenate([half, half[: k // 2][::-1]])
    extra = [pal, pal, np.concatenate([pal, pal[1:]]), np.full(k + 3, 7, np.uint32)]
    mins = np.concatenate([mins] + extra).astype(np.uint32)
    offs = np.concatenate([offs, offs[-1] + np.cumsum([len(x) for x in extra]).astype(np.uint64)])
    ctx.set_option("first_pass_mode", {"auto": 0, "one_table": 1, "partitioned": 2}[mode])
    try:
        for min_ab in (0, 2):
            t = ctx.kminmer_count_first(ctx.minimizers_from_host(mins, offs), k, min_ab)
            if mode == "partitioned":
                assert ctx.first_pass_info()["path"] == (2 if k <= 32 else 1)
            rec, vec = t.to_host()
            exp = orc.kminmer_count_first(mins, offs, k, min_ab)
            assert t.info()["n_solid"] == exp["n_solid"] and exp["n_solid"] > 0
            _assert_tables_equal(rec, vec, exp, k)
    finally:
        ctx.set_option("first_pass_mode", 0)


@pytest.mark.parametrize("form", ["slots", "slots_lazy", "slots_never_lazy", "slots_fused", "slots_two", "slots_round5", "slots_32_lanes", "buckets"])
@pytest.mark.parametrize("k", [13, 15, 17, 33, 65, 100])
def test_refined_and_index_deep_k_vs_oracle(ctx, orc, k, form):
    """The passes above firstK at deep k against the oracle on seeded inputs: previous table at k - 1 (+ a unitig overlay), refined count,
    index, the small-contig flags (unitigs shorter than k, CreateMdbg.hpp:1330-1352), the pass's table as the next pass's previous
    table -- with most reads shorter than k and unitigs much longer (what the reference's loop looks like from k = 38 on for 10 kb
    HiFi reads: the unitigs carry the table)."""
    rng = np.random.default_rng(1300 + k)
    genome, mins, offs = _genome_reads(rng, 6000, 500, int(1.6 * k) + 10)
    # unitigs = disjoint genome segments; eight of exactly k - 1 minimizers (the only length the small-contig branch can flag: no k-min-mer,
    # one (k-1)-min-mer), some of k - 2, k, k + 1, the rest long
    lens = [k - 1] * 8 + [k - 2, k, k + 1, 3, 1] + [int(x) for x in rng.integers(k, 6 * k, 12)]
    starts = np.concatenate([[0], np.cumsum(lens)])
    assert starts[-1] <= 6000
    ul = [genome[a:a + n] for a, n in zip(starts[:-1], lens)]
    uoffs = starts.astype(np.uint64)
    umins = np.concatenate(ul).astype(np.uint32)
    allm = np.concatenate([mins, umins]); alloff = np.concatenate([offs, offs[-1] + uoffs[1:]])
    prev_t = orc.kminmer_count_first(allm, alloff, k - 1, 0)
    prev_raw = orc.table_abundance_records(prev_t).tobytes()
    uab = np.concatenate([[0, 1, 2, 3, 4, 5, 2, 9], rng.integers(0, 6, len(lens) - 8)]).astype(np.uint32)
    oprev = orc.PrevAbundance(prev_raw)
    oprev.overlay_unitigs([(umins[int(uoffs[i]): int(uoffs[i + 1])], int(uab[i])) for i in range(len(uab)) if uab[i] != 4], k - 1)
    uab = np.where(uab == 4, 0xFFFFFFFF, uab).astype(np.uint32)
    d_reads = ctx.minimizers_from_host(mins, offs)
    d_unitigs = ctx.minimizers_from_host(umins, uoffs)
    dprev = ctx.prev_from_records(prev_raw)
    ctx.prev_overlay_unitigs(dprev, d_unitigs, uab, k - 1)
    ohi, olo, oab = oprev.arrays()
    assert np.array_equal(dprev.lookup(olo, ohi), oab)
    with _table_form(ctx, form):
        rec, vec = ctx.kminmer_count_refined(d_reads, d_unitigs, k, dprev).to_host()
        exp = orc.kminmer_count_refined(allm, alloff, k, oprev)
        assert exp["n"] > 0
        _assert_tables_equal(rec, vec, exp, k)
        t6 = ctx.kminmer_index(d_reads, d_unitigs, k, dprev)
        rec, vec = t6.to_host()
        exp = orc.kminmer_index(allm, alloff, k, oprev)
        assert exp["n"] > 0
        _assert_tables_equal(rec, vec, exp, k)
        exp1 = orc.kminmer_index(allm, alloff, k + 1, orc.PrevAbundance(orc.table_abundance_records(exp).tobytes()))
        rec, vec = ctx.kminmer_index(d_reads, d_unitigs, k + 1, t6).to_host()
        _assert_tables_equal(rec, vec, exp1, k + 1)
    flags = ctx.small_contigs(d_unitigs, k, k - 1, dprev)
    assert np.array_equal(flags, orc.small_contigs(umins, uoffs, k, k - 1, oprev)) and flags.any() and not flags.all()


def test_small_contigs_and_index_with_very_long_unitigs(ctx, orc):
    """Unitigs of more than 2^16 minimizers (a 16-lane group walks one for thousands of rounds; the instance offsets leave 32 bits of
    a sequence far behind) next to unitigs shorter than k - 1, k = 12 and 40: the index pass, the previous-abundance look-ups and the
    small-contig flags against the oracle."""
    rng = np.random.default_rng(4242)
    genome = rng.permutation(400_000).astype(np.uint32)
    lens = [70_000, 3, 150_000, 0, 11, 39, 40, 41, 66_000, 12, 100_000]
    ul, a = [], 0
    for n in lens:
        ul.append(genome[a:a + n][::-1] if n % 2 else genome[a:a + n]); a += n
    uoffs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    umins = np.concatenate(ul).astype(np.uint32)
    _, mins, offs = _genome_reads(rng, 400_000, 300, 90, genome=genome)
    allm = np.concatenate([mins, umins]); alloff = np.concatenate([offs, offs[-1] + uoffs[1:]])
    for k in (12, 40):
        prev_t = orc.kminmer_count_first(allm, alloff, k - 1, 0)
        prev_raw = orc.table_abundance_records(prev_t).tobytes()
        oprev = orc.PrevAbundance(prev_raw)
        uab = np.array([3, 2, 5, 0, 1, 2, 2, 2, 7, 0, 2], dtype=np.uint32)
        oprev.overlay_unitigs([(umins[int(uoffs[i]): int(uoffs[i + 1])], int(uab[i])) for i in range(len(uab))], k - 1)
        d_reads = ctx.minimizers_from_host(mins, offs)
        d_unitigs = ctx.minimizers_from_host(umins, uoffs)
        dprev = ctx.prev_from_records(prev_raw)
        ctx.prev_overlay_unitigs(dprev, d_unitigs, uab, k - 1)
        rec, vec = ctx.kminmer_index(d_reads, d_unitigs, k, dprev).to_host()
        exp = orc.kminmer_index(allm, alloff, k, oprev)
        assert exp["n"] > 300_000
        _assert_tables_equal(rec, vec, exp, k)
        flags = ctx.small_contigs(d_unitigs, k, k - 1, dprev)
        assert np.array_equal(flags, orc.small_contigs(umins, uoffs, k, k - 1, oprev))


import contextlib


@contextlib.contextmanager
def _table_form(ctx, form: str):
    """The passes above firstK in their forms (mdbg_set_option "index_table_form" / "refined_form" / "index_tuning"): "slots" -- one 32-byte
    slot per key with the library's default tuning --, "slots_fused" / "slots_two_kernels" -- look-up and insert in one kernel (the neighbour's
    abundance by a DPP move) or in two with an array in between, both with a slot's words in one trip and the insert's plain-load first look --,
    "slots_two" -- two windows of a lane in flight --, "slots_round4" -- the same tables with the kernels of rounds 1 - 4 --, "slots_round5" -- round 5's
    kernels, the pass's own table dropped and rebuilt from the rows --, "slots_pair_insert" / "slots_32_lanes" -- round 6's measured-and-not-taken variants --, "buckets" -- three keys per 64-byte sector,
    the refined pass by look-ups like an index pass (both measured, neither faster: DESIGN.md 4.2); "slots_lazy" / "slots_never_lazy" -- the index pass
    that inserts first and asks the previous table only where a key was not seen (index_lazy_kernel), always / never (the default decides by a sample)."""
    ctx.set_option("index_table_form", 0 if form == "buckets" else 1)
    ctx.set_option("refined_form", 0 if form == "buckets" else 1)
    # (bit 8 = 256: never the insert-first form, which the default otherwise chooses by a sample; bit 7 = 128: always)
    ctx.set_option("index_tuning", {"slots_round4": 0, "slots_two": 7, "slots_fused": 11, "slots_two_kernels": 3 | 256, "slots_round5": 3 | 256, "slots_pair_insert": 83,
                                    "slots_32_lanes": 115, "slots_lazy": 19 | 128, "slots_never_lazy": 19 | 256}.get(form, -1))
    ctx.set_option("keep_index_table", 0 if form == "slots_round5" else 1)
    try:
        yield
    finally:
        ctx.set_option("index_table_form", 1)
        ctx.set_option("refined_form", 1)
        ctx.set_option("index_tuning", -1)
        ctx.set_option("keep_index_table", 1)


@pytest.mark.parametrize("form", ["slots", "slots_lazy", "slots_never_lazy", "slots_fused", "slots_two_kernels", "slots_two", "slots_round4", "slots_round5", "slots_pair_insert", "slots_32_lanes", "buckets"])
@pytest.mark.parametrize("k", [5, 6, 9])
def test_refined_and_index_vs_oracle(ctx, orc, k, form):
    rng = np.random.default_rng(300 + k)
    # a "genome" of distinct minimizers; reads = random substrings in either orientation; unitigs =
    # disjoint genome segments (as in a compacted graph every k-min-mer belongs to ONE unitig, so the
    # overlay has no write conflicts -- the reference's own overlay is order-dependent otherwise)
    genome = rng.permutation(5000).astype(np.uint32)
    rl = []
    for _ in range(600):
        a = int(rng.integers(0, 4900)); n = int(rng.integers(0, 80))
        seg = genome[a:a + n]
        if rng.integers(0, 2): seg = seg[::-1]
        seg = seg.copy()
        if n and rng.integers(0, 4) == 0: seg[int(rng.integers(0, n))] = 6000 + int(rng.integers(0, 50))   # "errors"
        rl.append(seg)
    offs = np.concatenate([[0], np.cumsum([len(x) for x in rl])]).astype(np.uint64)
    mins = np.concatenate(rl).astype(np.uint32)
    cuts = np.sort(rng.choice(np.arange(1, 5000), 60, replace=False))
    ul = [genome[a:b] for a, b in zip(np.concatenate([[0], cuts]), np.concatenate([cuts, [5000]]))][:50]
    uoffs = np.concatenate([[0], np.cumsum([len(x) for x in ul])]).astype(np.uint64)
    umins = np.concatenate(ul).astype(np.uint32)
    # previous table = first-pass counts at k-1 over the reads (benchmark mode) + a unitig overlay
    prev_t = orc.kminmer_count_first(mins, offs, k - 1, 0)
    prev_raw = orc.table_abundance_records(prev_t).tobytes()
    uab = rng.integers(0, 5, len(uoffs) - 1).astype(np.uint32)
    oprev = orc.PrevAbundance(prev_raw)
    # 4 stands for "unitig without a refined abundance" (skipped); 0 and 1 are real values
    oprev.overlay_unitigs([(umins[int(uoffs[i]): int(uoffs[i + 1])], int(uab[i])) for i in range(len(uab)) if uab[i] != 4], k - 1)
    uab = np.where(uab == 4, 0xFFFFFFFF, uab).astype(np.uint32)
    d_reads = ctx.minimizers_from_host(mins, offs)
    d_unitigs = ctx.minimizers_from_host(umins, uoffs)
    dprev = ctx.prev_from_records(prev_raw)
    ctx.prev_overlay_unitigs(dprev, d_unitigs, uab, k - 1)
    # lookups agree with the oracle map
    ohi, olo, oab = oprev.arrays()
    assert np.array_equal(dprev.lookup(olo, ohi), oab)
    allm = np.concatenate([mins, umins]); alloff = np.concatenate([offs, offs[-1] + uoffs[1:]])
    with _table_form(ctx, form):
        t5 = ctx.kminmer_count_refined(d_reads, d_unitigs, k, dprev)
        rec, vec = t5.to_host()
        _assert_tables_equal(rec, vec, orc.kminmer_count_refined(allm, alloff, k, oprev), k)
        t6 = ctx.kminmer_index(d_reads, d_unitigs, k, dprev)
        rec, vec = t6.to_host()
        assert vec is None
        _assert_tables_equal(rec, vec, orc.kminmer_index(allm, alloff, k, oprev), k)
        # the table a pass hands out serves the next as its previous table (the bucket form leaves its own table behind as the image)
        exp = orc.kminmer_index(allm, alloff, k + 1, orc.PrevAbundance(orc.table_abundance_records(orc.kminmer_index(allm, alloff, k, oprev)).tobytes()))
        rec, vec = ctx.kminmer_index(d_reads, d_unitigs, k + 1, t6).to_host()
        _assert_tables_equal(rec, vec, exp, k + 1)
        # ... and answers look-ups like any table
        r6 = t6.to_host()[0]
        if len(r6):
            assert np.array_equal(t6.lookup(r6["lo"].astype(np.uint64), r6["hi"].astype(np.uint64)), r6["abundance"])


@pytest.mark.parametrize("n_ranks", [2, 3, 8])
def test_sharded_first_pass_on_one_gpu(ctx, orc, n_ranks):
    """All kernels of the sharded first pass, with the two all-to-alls emulated on one device: shard the
    reads, mdbg_shard_begin per shard, concatenate the rows per owner, mdbg_shard_reduce per owner, route the
    replies back to the senders, mdbg_shard_finish per shard; the union of the per-rank tables must equal the
    single-GPU table, and every row's global count the count over all reads."""
    import ctypes as C
    k = 4
    rng = np.random.default_rng(11 + n_ranks)
    mins, offs = _random_minimizer_reads(rng, 600, 30)
    n_reads = len(offs) - 1
    cuts = np.linspace(0, n_reads, n_ranks + 1).astype(int)
    shards = []
    for r in range(n_ranks):
        so = offs[cuts[r]: cuts[r + 1] + 1] - offs[cuts[r]]
        sm = mins[int(offs[cuts[r]]): int(offs[cuts[r + 1]])]
        shards.append(ctx.minimizers_from_host(sm, so))
    hip = C.CDLL("libamdhip64.so.7")   # already loaded by libmdbg_hip.so (same SONAME)

    def to_host(ptr, shape):
        out = np.zeros(shape, dtype=np.uint64)
        if out.nbytes:
            assert hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(out.nbytes), 2) == 0
        return out

    def to_device(a):
        buf = C.c_void_p()
        assert hip.hipMalloc(C.byref(buf), C.c_size_t(max(a.nbytes, 8))) == 0
        if a.nbytes:
            assert hip.hipMemcpy(buf, a.ctypes.data_as(C.c_void_p), C.c_size_t(a.nbytes), 1) == 0
        return buf

    # begin: rows destined to each owner (host copies stand in for the all-to-all)
    sh = [ctx.shard_begin(shards[r], k, n_ranks) for r in range(n_ranks)]
    rw = sh[0].row_words
    sent = [to_host(s.d_rows, (s.n_rows, rw)) for s in sh]
    per_owner = [[] for _ in range(n_ranks)]
    for r in range(n_ranks):
        o = 0
        for dst in range(n_ranks):
            c = int(sh[r].counts[dst])
            per_owner[dst].append(sent[r][o: o + c]); o += c
    # reduce on the owners; replies go back split by source rank
    recv_bufs, replies = [], []
    for dst in range(n_ranks):
        rows = np.ascontiguousarray(np.concatenate(per_owner[dst]))
        buf = to_device(rows)
        recv_bufs.append(buf)
        replies.append(to_host(sh[dst].reduce(buf.value, len(rows)), (len(rows),)))
    # exact global counts, from the oracle over all the reads
    exp = orc.kminmer_count_first(mins, offs, k, 0)
    recs, vecs, n_solid = [], [], 0
    for r in range(n_ranks):
        glob = []
        for dst in range(n_ranks):
            o = sum(int(sh[src].counts[dst]) for src in range(r))
            glob.append(replies[dst][o: o + int(sh[r].counts[dst])])
        glob = np.ascontiguousarray(np.concatenate(glob))
        assert len(glob) == sh[r].n_rows
        gbuf = to_device(glob)
        t = sh[r].finish(gbuf.value, 0)
        hip.hipFree(gbuf)
        rec, vec = t.to_host()
        recs.append(rec); vecs.append(vec); n_solid += t.info()["n_solid"]
        t.free()
    for s, buf in zip(sh, recv_bufs):
        s.free()
        hip.hipFree(buf)
    assert n_solid == exp["n_solid"]
    _assert_tables_equal(np.concatenate(recs), np.concatenate(vecs), exp, k)


# ---- qualities (FASTQ) and invalid characters (N) -------------------------------------------------
@pytest.mark.parametrize("tag,hpc", [("fastq_hpc_k15", True), ("fastq_nohpc_k15", False)])
def test_scan_edge_reads_fastq_golden(ctx, tag, hpc):
    seqs, quals = H.read_fastq(os.path.join(H.GOLDEN, "edge", "edge.fastq"))
    seqs = [s.upper() for s in seqs]
    rep = np.frombuffer(H.golden_bytes("edge", f"repetitiveMinimizers.{tag}.bin"), "<u4")
    reads = ctx.reads_from_ascii(seqs, quals)
    h = ctx.scan(reads, K=15, density=0.005, hpc=hpc, repetitive=rep).to_host()
    assert formats.build_read_data_init(h) == H.golden_bytes("edge", f"read_data_init.{tag}.txt")


def test_ont_100_golden_end_to_end(ctx, orc):
    m = H.load_manifest("ont_100")
    spec = H.spec_from_manifest(m)
    reads = ctx.reads_synthetic(spec)                      # bases + qualities generated in HBM
    seqs, quals = H.regenerate_reads(m)
    b, q = reads.get(42, with_quality=True)
    assert b == seqs[42] and q == quals[42]
    # repetitive-minimizer pre-pass (ReadSelection.hpp:497-561): bare parse at the correction density
    pre = ctx.scan(reads, K=m["K"], density=0.025, hpc=False, apply_read_filters=False)
    rep_gpu = ctx.repetitive_minimizers(pre)
    rep = np.frombuffer(H.golden_bytes("ont_100", "repetitiveMinimizers.bin"), "<u4")
    assert len(rep_gpu) == len(rep)                        # which of the tied top minimizers is unstable in the reference
    mins = ctx.scan(reads, K=m["K"], density=m["density"], hpc=False, repetitive=rep)
    h = mins.to_host()
    assert formats.build_read_data_init(h) == H.golden_bytes("ont_100", "read_data_init.txt")
    st = formats.parse_read_stats(H.golden_bytes("ont_100", "read_stats.txt"))
    last_k = orc.lib().orc_compute_last_k(m["density"], st["n50"], 4, 0)
    corr = ctx.purge_palindromes(mins, 4, last_k)
    hc = corr.to_host(full=False)
    assert formats.write_minimizer_reads(hc["minimizers"], hc["offsets"]) == H.golden_bytes("ont_100", "read_data_corrected.txt")
    rec, vec = ctx.kminmer_count_first(corr, m["k"], m["min_abundance"]).to_host()
    exp_ab = np.fromfile(os.path.join(H.GOLDEN, "ont_100", "kminmerData_abundance.sorted.bin"), formats.ABUNDANCE_DTYPE)
    assert np.array_equal(formats.sorted_abundance_records(rec), exp_ab)
    exp_v = np.fromfile(os.path.join(H.GOLDEN, "ont_100", "kminmerData_min.sorted.bin"), "<u4").reshape(-1, m["k"])
    assert np.array_equal(formats.sorted_vector_records(vec.astype("<u4").tobytes(), m["k"]), exp_v)


@pytest.mark.parametrize("hpc", [True, False])
def test_scan_with_qualities_vs_oracle(ctx, orc, hpc):
    rng = np.random.default_rng(77 + int(hpc))
    lens = list(rng.integers(1, 300, 30)) + list(rng.integers(300, 12000, 40)) + [2048, 2049, 4096, 6144 + 15]
    seqs, quals = [], []
    for n in lens:
        n = int(n)
        s = synth.CODE2ASCII[rng.integers(0, 4, n)]
        if rng.integers(0, 3) == 0 and n > 50:                    # homopolymer stretches across tile borders
            a = int(rng.integers(0, n - 40)); s[a: a + int(rng.integers(5, 40))] = s[a]
        seqs.append(bytes(s))
        qmax = 94 if len(seqs) % 2 else 12                        # every other read is low quality
        quals.append(bytes((rng.integers(0, qmax, n) + 33).astype(np.uint8)))
    _check_scan_against_oracle(ctx, orc, seqs, quals, 15, 0.02, hpc)
    # quality filter (ReadSelection.hpp:901-909)
    reads = ctx.reads_from_ascii(seqs, quals)
    h = ctx.scan(reads, K=15, density=0.02, hpc=hpc, min_read_quality=9.0).to_host()
    exp = b"".join(orc.read_selection(s, q, K=15, density=0.02, hpc=hpc, min_read_quality=9.0)["record"] for s, q in zip(seqs, quals))
    assert formats.build_read_data_init(h) == exp
    assert (h["flags"] & 2).any() and not (h["flags"] & 2).all()


def test_scan_reads_with_n(ctx, orc):
    """Reference outputs (refdrv fn_scan) for reads holding N / n, mixed case, soft-masked blocks and IUPAC letters: the
    characters go to the device as they are (EncoderRLE compares characters, Commons.hpp:4177-4178)."""
    with open(os.path.join(H.GOLDEN, "fn", "fn_golden.json")) as f:
        gold = json.load(f)
    n_mixed = 0
    for section in ("scan_n", "scan_case"):
        for key, gg in gold[section].items():
            seqs = [s.encode() for s in gg["inputs"]]
            reads = ctx.reads_from_ascii(seqs)
            h = ctx.scan(reads, K=gg["K"], density=gg["density"], hpc=bool(gg["hpc"]), apply_read_filters=False).to_host()
            for i, out in enumerate(gg["outputs"]):
                toks = out.split()
                exp = [tuple(int(x) for x in t.split(":")) for t in toks[2:]]
                a, b = int(h["offsets"][i]), int(h["offsets"][i + 1])
                got = list(zip(h["minimizers"][a:b].tolist(), h["pos"][a:b].tolist(), h["dir"][a:b].tolist()))
                assert got == exp, (section, key, i)
                n_mixed += gg["inputs"][i] not in (gg["inputs"][i].upper(), gg["inputs"][i].lower())
    assert n_mixed >= 40
    # mixed case with the read filters on, against the oracle (qualities too: the coordinate map follows the runs)
    rng0 = np.random.default_rng(11)
    seqs, quals = [], []
    for n in list(rng0.integers(20, 400, 12)) + list(rng0.integers(400, 9000, 12)):
        s = bytearray(np.repeat(synth.CODE2ASCII[rng0.integers(0, 4, int(n))], rng0.choice([1, 1, 2, 5], int(n))).tobytes())
        for _ in range(int(rng0.integers(1, 8))):
            a = int(rng0.integers(0, len(s))); b = min(len(s), a + int(rng0.integers(1, 120)))
            s[a:b] = bytes(s[a:b]).lower()
        seqs.append(bytes(s))
        quals.append(bytes((rng0.integers(0, 60, len(s)) + 33).astype(np.uint8)))
    for hpc in (True, False):
        _check_scan_against_oracle(ctx, orc, seqs, None, 15, 0.02, hpc)
        _check_scan_against_oracle(ctx, orc, seqs, quals, 15, 0.02, hpc)
    # random reads with N against the oracle, filters on (N counts with its 2-bit code in the complexity score)
    rng = np.random.default_rng(5)
    seqs = []
    for n in list(rng.integers(20, 400, 20)) + list(rng.integers(400, 9000, 20)):
        s = synth.CODE2ASCII[rng.integers(0, 4, int(n))]
        for _ in range(int(rng.integers(0, 6))):
            a = int(rng.integers(0, n)); s[a: a + int(rng.integers(1, 5))] = ord("N")
        seqs.append(bytes(s))
    for hpc in (True, False):
        _check_scan_against_oracle(ctx, orc, seqs, None, 15, 0.02, hpc)


def test_scan_without_end_trim(ctx):
    """N4: GenerateGfa's unitig scan sets MinimizerParser::_trimBps = 0 (GenerateGfa.hpp:366)."""
    with open(os.path.join(H.GOLDEN, "fn", "fn_golden.json")) as f:
        g = json.load(f)
    # scan_notrim_block: compressed lengths 2048*b + K + {-1, 0, 1} at a density that keeps the reads in the block kernel's
    # stage -- at exactly 2048*b + K bases the tail of the round-2 kernel was left with 2049 positions and lost the verdicts
    # of the first four of every lane
    for key, gg in list(g["scan_notrim"].items()) + list(g["scan_notrim_block"].items()):
        reads = ctx.reads_from_ascii([s.encode() for s in gg["inputs"]])
        h = ctx.scan(reads, K=gg["K"], density=gg["density"], hpc=bool(gg["hpc"]), apply_read_filters=False, no_end_trim=True).to_host()
        for i, out in enumerate(gg["outputs"]):
            exp = [tuple(int(x) for x in t.split(":")) for t in out.split()[2:]]
            a, b = int(h["offsets"][i]), int(h["offsets"][i + 1])
            got = list(zip(h["minimizers"][a:b].tolist(), h["pos"][a:b].tolist(), h["dir"][a:b].tolist()))
            assert got == exp, (key, i)


def test_scan_block_border_lengths_vs_oracle(ctx, orc):
    """Every compressed length around the block kernel's borders (2048*b + K - 2 ... + 2), with and without the end trim, HPC on
    and off, K = 11 / 15 / 16, against the oracle (itself pinned on these lengths by fn_golden's scan_notrim_block)."""
    rng = np.random.default_rng(77)

    def seq_of(n, hpc):
        c = np.empty(n, dtype=np.int64)
        c[0] = rng.integers(0, 4); c[1:] = rng.integers(1, 4, n - 1)
        c = np.cumsum(c) % 4
        if hpc:
            c = np.repeat(c, rng.choice([1, 1, 2, 3], n))
        return bytes(synth.CODE2ASCII[c])

    for hpc in (True, False):
        for K in (11, 15, 16):
            seqs = [seq_of(2048 * b + K + d, hpc) for b in (1, 2, 4) for d in (-2, -1, 0, 1, 2)]
            for trim in (False, True):
                reads = ctx.reads_from_ascii(seqs)
                h = ctx.scan(reads, K=K, density=0.04, hpc=hpc, apply_read_filters=False, no_end_trim=not trim).to_host()
                for i, s in enumerate(seqs):
                    e = orc.minimizer_parse(s, K, 0.04, hpc, trim=1 if trim else 0)
                    a, b = int(h["offsets"][i]), int(h["offsets"][i + 1])
                    assert (h["minimizers"][a:b].tolist(), h["pos"][a:b].tolist(), h["dir"][a:b].tolist()) == \
                        (list(e[0]), list(e[1]), list(e[2])), (hpc, K, trim, i, len(s))


def test_correction_scan_and_density_threshold(ctx, orc):
    """N1: the correction-density scan (ReadCorrection::ReadSelectionFunctor: no read filters, inclusive quality
    span) and Utils::applyDensityThreshold, against the reference's own outputs (fn_golden.json) and the oracle."""
    with open(os.path.join(H.GOLDEN, "fn", "fn_golden.json")) as f:
        g = json.load(f)
    for key, gg in g["corrscan"].items():
        reads = ctx.reads_from_ascii([r.encode() for r in gg["reads"]], [q.encode() for q in gg["quals"]])
        m = ctx.scan(reads, K=gg["K"], density=gg["density"], hpc=bool(gg["hpc"]), apply_read_filters=False, quality_window=1)
        h = m.to_host()
        for i, out in enumerate(gg["outputs"]):
            exp = [tuple(int(x) for x in t.split(":")) for t in out.split()[1:]]
            a, b = int(h["offsets"][i]), int(h["offsets"][i + 1])
            got = list(zip(h["minimizers"][a:b].tolist(), h["pos"][a:b].tolist(), h["dir"][a:b].tolist(), h["qual"][a:b].tolist()))
            assert got == exp, (key, i)
        # down-sample the same reads to the assembly density: every per-minimizer array follows
        low = ctx.apply_density_threshold(m, 0.005).to_host()
        for i in range(len(gg["reads"])):
            a, b = int(h["offsets"][i]), int(h["offsets"][i + 1])
            keep = orc.apply_density_threshold(h["minimizers"][a:b], 0.005) + a
            c, d = int(low["offsets"][i]), int(low["offsets"][i + 1])
            for f_ in ("minimizers", "pos", "dir", "qual"):
                assert low[f_][c:d].tolist() == h[f_][keep].tolist(), (key, i, f_)
        assert low["read_length"].tolist() == h["read_length"].tolist()
    for dens, gg in g["density"].items():
        lists = [np.array(line.split(), dtype=np.uint64).astype(np.uint32) for line in gg["inputs"]]
        offs = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.uint64)
        m = ctx.minimizers_from_host(np.concatenate(lists), offs)
        low = ctx.apply_density_threshold(m, float(dens)).to_host(full=False)
        for i, out in enumerate(gg["outputs"]):
            exp = lists[i][[int(x) for x in out.split()[1:]]]
            assert low["minimizers"][int(low["offsets"][i]): int(low["offsets"][i + 1])].tolist() == exp.tolist(), (dens, i)
    # large random case against the oracle
    rng = np.random.default_rng(9)
    mins = rng.integers(0, 2**32, 300000, dtype=np.uint64).astype(np.uint32)
    offs = np.concatenate([[0], np.sort(rng.integers(0, len(mins), 999)), [len(mins)]]).astype(np.uint64)
    low = ctx.apply_density_threshold(ctx.minimizers_from_host(mins, offs), 0.2).to_host(full=False)
    keep = orc.apply_density_threshold(mins, 0.2)
    assert low["minimizers"].tolist() == mins[keep].tolist()
    assert low["offsets"].tolist() == np.searchsorted(keep, offs).tolist()


def test_multi_k_loop_benchmark_mode(ctx, orc):
    """BASELINE.json configs[2] at test scale: k = 4 .. 11 over the same minimizer-space reads, reads only,
    previous table = own k-1 output (SURVEY.md 8(d) "benchmark mode"); every k must equal the oracle's loop."""
    spec = synth.hifi_spec(400, seed=9, read_len=8000, coverage=30.0)
    reads = ctx.reads_synthetic(spec)
    mins = ctx.scan(reads, K=15, density=0.005, hpc=True)
    corr = ctx.purge_palindromes(mins, 4, 100)
    hc = corr.to_host(full=False)
    m, o = hc["minimizers"], hc["offsets"]
    t_gpu = ctx.kminmer_count_first(corr, 4, 0)
    t_orc = orc.kminmer_count_first(m, o, 4, 0)
    rec, vec = t_gpu.to_host()
    _assert_tables_equal(rec, vec, t_orc, 4)
    for k in range(5, 12):
        prev_raw = rec.tobytes()
        oprev = orc.PrevAbundance(prev_raw)
        oprev.overlay_unitigs([], k - 1)
        if k == 5:
            t_gpu2 = ctx.kminmer_count_refined(corr, None, k, t_gpu)
            t_orc = orc.kminmer_count_refined(m, o, k, oprev)
        else:
            t_gpu2 = ctx.kminmer_index(corr, None, k, t_gpu)
            t_orc = orc.kminmer_index(m, o, k, oprev)
        rec, vec = t_gpu2.to_host()
        assert len(rec) > 0
        _assert_tables_equal(rec, vec, t_orc, k)
        t_gpu = t_gpu2


def test_ont_like_reads_vs_oracle(ctx, orc):
    """BASELINE.json configs[3] at test scale: 20 kb reads, 2 % errors, qualities, no HPC, nanoMDBG densities,
    repetitive filter from the 0.025 pre-pass."""
    spec = synth.SynthSpec(n_reads=300, read_len=20_000, seed=31, sub_rate=0.02, species_len=[150_000, 90_000],
                           species_weight=[0.6, 0.4], with_quality=True, name="ont")
    reads = ctx.reads_synthetic(spec)
    pre = ctx.scan(reads, K=15, density=0.025, hpc=False, apply_read_filters=False)
    rep = ctx.repetitive_minimizers(pre)
    assert len(rep) >= 1
    asc = synth.codes_to_ascii(synth.read_codes(spec, 0, spec.n_reads))
    q = synth.read_qualities(spec, 0, spec.n_reads)
    h = ctx.scan(reads, K=15, density=0.005, hpc=False, repetitive=rep).to_host()
    exp = b"".join(orc.read_selection(asc[r].tobytes(), q[r].tobytes(), K=15, density=0.005, hpc=False, repetitive=rep)["record"]
                   for r in range(spec.n_reads))
    assert formats.build_read_data_init(h) == exp


@pytest.mark.parametrize("name", ["hifi_200", "ont_100"])
def test_edge_index_vs_reference_log(ctx, orc, name):
    """EdgeIndexer on the device against the edge count and checksum the reference logs, and
    against the oracle's edge set."""
    m = H.load_manifest(name)
    raw = H.golden_bytes(name, "read_data_corrected.txt")
    mins, offs = formats.parse_minimizer_reads(raw)
    table = ctx.kminmer_count_first(ctx.minimizers_from_host(mins, offs), m["k"], m["min_abundance"])
    edges, ck = ctx.edge_index(table)
    log = m["reference_log"]
    assert edges.info()["n_records"] == log["n_edges"] and ck == log["edge_checksum"]
    rec, vec = table.to_host()
    ohi, olo, ock = orc.edge_index(vec)
    keys = edges.keys_to_host()
    order = np.lexsort((keys[:, 0], keys[:, 1]))
    assert np.array_equal(keys[order, 0], olo) and np.array_equal(keys[order, 1], ohi) and ock == ck


def test_full_size_properties(ctx, orc):
    """BASELINE.json configs[1] at FULL size (1 M x 10 kb HiFi reads, the bench workload), checked through
    properties that do not need the oracle to run over 10 Gbp:
      * the oracle on a sample of the batch (first / last reads) equals the corresponding slice of the full scan;
      * shard invariance: scanning the two halves separately gives the two halves of the full scan;
      * purge is idempotent;
      * linearity of the counts: the sharded first pass over the two halves (both exchanges emulated on this
        device) yields, as a multiset, exactly the table of the single call over the whole batch;
      * conservation: the abundances of the solid rows account for every instance of a solid k-min-mer."""
    import ctypes as C
    n, L, k = 1_000_000, 10_000, 4
    spec = synth.hifi_spec(n, seed=42, read_len=L, coverage=50.0)
    reads = ctx.reads_synthetic(spec)
    full = ctx.scan(reads, K=15, density=0.005, hpc=True)
    h = full.to_host()
    assert int(h["offsets"][-1]) == len(h["minimizers"]) and (np.diff(h["offsets"].astype(np.int64)) >= 0).all()
    # oracle on a sample
    for first in (0, n - 300):
        bases, offs = reads.export_ascii(first, 300)
        for r in range(0, 300, 7):
            seq = bases[int(offs[r]): int(offs[r + 1])].tobytes()
            o = orc.read_selection(seq, None, K=15, density=0.005, hpc=True)
            a, b = int(h["offsets"][first + r]), int(h["offsets"][first + r + 1])
            assert h["minimizers"][a:b].tolist() == o["minimizers"].tolist()
            assert h["pos"][a:b].tolist() == o["pos"].tolist() and h["dir"][a:b].tolist() == o["dir"].tolist()
    reads.free()
    # shard invariance of the scan (the halves are regenerated: same generator, other read range)
    halves = []
    cut = int(h["offsets"][n // 2])
    for i, (f, lo, hi) in enumerate(((0, 0, cut), (n // 2, cut, len(h["minimizers"])))):
        part = ctx.reads_synthetic(spec, first_read=f, n_reads=n // 2)
        m = ctx.scan(part, K=15, density=0.005, hpc=True)
        part.free()
        hp = m.to_host(full=False)
        assert np.array_equal(hp["minimizers"], h["minimizers"][lo:hi])
        assert np.array_equal(hp["offsets"], h["offsets"][f: f + n // 2 + 1] - h["offsets"][f])
        halves.append(m)
    # purge: idempotent
    corr = ctx.purge_palindromes(full, 4, 100)
    corr2 = ctx.purge_palindromes(corr, 4, 100)
    hc, hc2 = corr.to_host(full=False), corr2.to_host(full=False)
    assert np.array_equal(hc["minimizers"], hc2["minimizers"]) and np.array_equal(hc["offsets"], hc2["offsets"])
    corr2.free()
    # single call over the whole batch
    t = ctx.kminmer_count_first(corr, k, 0)
    rec, vec = t.to_host()
    info = t.info()
    n_inst = int(np.maximum(np.diff(hc["offsets"].astype(np.int64)) - (k - 1), 0).sum())
    solid = rec[: info["n_solid"]]
    assert (solid["abundance"] > 1).all() and (rec[info["n_solid"]:]["abundance"] == 1).all()
    assert int(solid["abundance"].sum()) <= n_inst
    # sharded over the halves, exchanges emulated
    hip = C.CDLL("libamdhip64.so.7")

    def to_host(ptr, shape):
        out = np.zeros(shape, dtype=np.uint64)
        if out.nbytes:
            assert hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(out.nbytes), 2) == 0
        return out

    def to_device(a):
        buf = C.c_void_p()
        assert hip.hipMalloc(C.byref(buf), C.c_size_t(max(a.nbytes, 8))) == 0
        if a.nbytes:
            assert hip.hipMemcpy(buf, a.ctypes.data_as(C.c_void_p), C.c_size_t(a.nbytes), 1) == 0
        return buf

    shards = [ctx.purge_palindromes(m, 4, 100) for m in halves]
    sh = [ctx.shard_begin(s, k, 2) for s in shards]
    rw = sh[0].row_words
    sent = [to_host(s.d_rows, (s.n_rows, rw)) for s in sh]
    replies = []
    for dst in range(2):
        parts = []
        for src in range(2):
            o = int(sh[src].counts[:dst].sum())
            parts.append(sent[src][o: o + int(sh[src].counts[dst])])
        rows = np.ascontiguousarray(np.concatenate(parts))
        buf = to_device(rows)
        replies.append(to_host(sh[dst].reduce(buf.value, len(rows)), (len(rows),)))
        hip.hipFree(buf)
    recs, vecs, n_solid = [], [], 0
    for r in range(2):
        glob = np.ascontiguousarray(np.concatenate([
            replies[dst][sum(int(sh[src].counts[dst]) for src in range(r)):][: int(sh[r].counts[dst])] for dst in range(2)]))
        gbuf = to_device(glob)
        ts = sh[r].finish(gbuf.value, 0)
        hip.hipFree(gbuf)
        rs, vs = ts.to_host()
        recs.append(rs); vecs.append(vs); n_solid += ts.info()["n_solid"]
        ts.free(); sh[r].free()
    assert n_solid == info["n_solid"]
    assert np.array_equal(formats.sorted_abundance_records(np.concatenate(recs)), formats.sorted_abundance_records(rec))
    assert np.array_equal(formats.sorted_vector_records(np.concatenate(vecs).astype("<u4").tobytes(), k),
                          formats.sorted_vector_records(vec.astype("<u4").tobytes(), k))


# ---- empty, ragged and extreme inputs ---------------------------------------------------------------
def test_empty_and_degenerate_batches(ctx, orc):
    """No reads, empty reads, reads shorter than l, reads of exactly l / l+1 / l+2 bases (0 / 0 / 1 eligible l-mers
    with the end trim), one-base homopolymers -- alone and mixed with normal reads."""
    rng = np.random.default_rng(77)
    rnd = lambda n: bytes(synth.CODE2ASCII[rng.integers(0, 4, n)])
    # empty batch through the whole chain
    reads = ctx.reads_from_ascii([])
    m = ctx.scan(reads, K=15, density=0.005, hpc=True)
    assert m.info() == dict(n_reads=0, n_minimizers=0)
    c = ctx.purge_palindromes(m, 4, 100)
    t = ctx.kminmer_count_first(c, 4, 0)
    assert t.info()["n_records"] == 0
    low = ctx.apply_density_threshold(m, 0.001)
    assert low.info()["n_minimizers"] == 0
    sh = ctx.shard_begin(c, 4, 3)
    assert sh.n_rows == 0
    sh.reduce(0, 0)
    assert sh.finish(0, 0).info()["n_records"] == 0
    sh.free()
    # degenerate reads
    seqs = [b"", b"A", b"ACGT", rnd(14), rnd(15), rnd(16), rnd(17), b"A" * 5000, b"AC" * 3000, rnd(3000), b"", rnd(40), b"T" * 15, rnd(2048),
            rnd(2047), rnd(2049), rnd(2048 + 15), rnd(4096), b"G"]
    for hpc in (True, False):
        for K, dens in ((15, 0.3), (16, 0.9), (11, 0.05)):
            _check_scan_against_oracle(ctx, orc, seqs, None, K, dens, hpc)
    # reads with fewer than k minimizers contribute no instance; k larger than any read
    mins = ctx.scan(ctx.reads_from_ascii(seqs), K=15, density=0.3, hpc=True)
    corr = ctx.purge_palindromes(mins, 4, 100)
    hc = corr.to_host(full=False)
    for k in (2, 4, 50, 5000):
        t = ctx.kminmer_count_first(corr, k, 0)
        exp = orc.kminmer_count_first(hc["minimizers"], hc["offsets"], k, 0)
        rec, vec = t.to_host()
        assert t.info()["n_solid"] == exp["n_solid"] and len(rec) == exp["n"]
        if len(rec):
            _assert_tables_equal(rec, vec, exp, k)


def test_extreme_reads(ctx, orc):
    """One 3 Mbp read next to short ones (1500 tiles in one wave), a density that overflows every padded output slot
    (the re-run path), and the largest minimizer size (l = 16: full 32-bit minimizers)."""
    rng = np.random.default_rng(78)
    rnd = lambda n: bytes(synth.CODE2ASCII[rng.integers(0, 4, n)])
    seqs = [rnd(200), rnd(3_000_000), rnd(50), rnd(100_000)]
    for hpc in (True, False):
        _check_scan_against_oracle(ctx, orc, seqs, None, 16, 0.005, hpc)
    _check_scan_against_oracle(ctx, orc, [rnd(20000), rnd(64), rnd(9000)], None, 15, 0.99, True)    # nearly every position selected
    _check_scan_against_oracle(ctx, orc, [rnd(20000), rnd(64), rnd(9000)], None, 8, 1.0, False)     # density 1: threshold saturates


def test_packed_upload_with_qualities_equals_ascii(ctx):
    """mdbg_reads_from_packed + mdbg_reads_attach_qualities (what the host feed sends) == mdbg_reads_from_ascii."""
    rng = np.random.default_rng(21)
    codes = [rng.integers(0, 4, int(n)).astype(np.uint8) for n in (1, 63, 64, 65, 700, 5000, 31, 2048)]
    seqs = [bytes(synth.CODE2ASCII[c]) for c in codes]
    quals = [bytes((rng.integers(0, 50, len(s)) + 33).astype(np.uint8)) for s in seqs]
    words, woff, lens = synth.pack_reads(codes)
    a = ctx.scan(ctx.reads_from_ascii(seqs, quals), K=13, density=0.05, hpc=True).to_host()
    b = ctx.scan(ctx.reads_from_packed(words, woff, lens, quals), K=13, density=0.05, hpc=True).to_host()
    for key in a:
        assert np.array_equal(a[key], b[key]) or (key == "mean_quality" and _nan_eq(a[key], b[key])), key
    assert formats.build_read_data_init(a) == formats.build_read_data_init(b)
    from metamdbg_amd import capi
    with pytest.raises(capi.MdbgError):      # one quality per base
        ctx.reads_from_packed(words, woff, lens, [q[:-1] if len(q) > 5 else q for q in quals])


@pytest.mark.parametrize("name", ["hifi_200", "ont_100"])
def test_unitig_edge_index_vs_oracle_and_reference_log(ctx, orc, name):
    """N2, second half: UnitigEdgeIndexer on the reference's own unitigs; plus random sequences incl. ones shorter
    than k, exactly k long (first == last k-min-mer) and palindromic ones."""
    m = H.load_manifest(name)
    k = m["k"]
    mins, offs, _ = formats.parse_unitig_nodes(H.golden_bytes(name, "unitigGraph.nodes.bin"))
    t, ck = ctx.unitig_edge_index(ctx.minimizers_from_host(mins, offs), k)
    assert t.info()["n_records"] == m["reference_log"]["n_unitig_edges"]
    hi, lo, ock = orc.unitig_edge_index(mins, offs, k)
    keys = t.keys_to_host()
    got = keys[np.lexsort((keys[:, 0], keys[:, 1]))]
    assert np.array_equal(got[:, 1], hi) and np.array_equal(got[:, 0], lo) and ck == ock
    rng = np.random.default_rng(5)
    for kk in (3, 4, 7):
        lens = np.concatenate([rng.integers(0, kk + 3, 300), rng.integers(kk, 60, 300)])
        o = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        mm = rng.integers(0, 6, int(o[-1])).astype(np.uint32)            # tiny alphabet: ties and palindromes
        t, ck = ctx.unitig_edge_index(ctx.minimizers_from_host(mm, o), kk)
        hi, lo, ock = orc.unitig_edge_index(mm, o, kk)
        keys = t.keys_to_host()
        got = keys[np.lexsort((keys[:, 0], keys[:, 1]))] if len(keys) else keys
        assert len(got) == len(hi) and np.array_equal(got[:, 1], hi) and np.array_equal(got[:, 0], lo) and ck == ock


def _multik_cases():
    from tests import multik_fixture as mk
    return [(s, k) for s in mk.SETS + mk.DEEP_SETS for k in mk.steps(s)]


@pytest.mark.parametrize("form", ["slots", "slots_lazy", "slots_never_lazy", "slots_fused", "slots_two_kernels", "slots_round4", "buckets"])
@pytest.mark.parametrize("name,k", _multik_cases())
def test_next_k_tables_equal_reference_multik(ctx, name, k, form):
    """Rows A13 / A14 against the REFERENCE: previous table + unitig overlay, refined count (k = firstK+1), index
    (k >= firstK+2) and the small-contig branch, on the inputs the reference's `graph` read in its own multi-k loop and
    the tables it wrote (tests/golden/*_multik: k = 5 .. 11 with --max-k 11; tests/golden/*_deepk: the DEFAULT loop, no --max-k,
    lastK = N50 x density x 2 (Commons.hpp:1726-1741) = 100 / 200, at k = 12 .. 16, 24, 31 .. 33, 48, 64, 65, 100, lastK -- the generic
    window hash with Murmur tails of 0 .. 3 words, k above the partitioned pass's 32, k above every read's minimizer count)."""
    from tests import multik_fixture as mk
    fx = mk.load(name, k)
    P = fx["params"]
    (rm, ro), (um, uo) = fx["reads"], fx["unitigs"]
    d_reads = ctx.minimizers_from_host(rm, ro)
    d_unitigs = ctx.minimizers_from_host(um, uo)
    dprev = ctx.prev_from_records(fx["prev_records"])
    if fx["prev_unitigs"]:
        pm = np.concatenate([u for u, _ in fx["prev_unitigs"]]).astype(np.uint32)
        po = np.concatenate([[0], np.cumsum([len(u) for u, _ in fx["prev_unitigs"]])]).astype(np.uint64)
        pa = np.array([a for _, a in fx["prev_unitigs"]], dtype=np.uint32)
        ctx.prev_overlay_unitigs(dprev, ctx.minimizers_from_host(pm, po), pa, P.prev_k)
    with _table_form(ctx, form):
        t = (ctx.kminmer_count_refined if k == P.first_k + 1 else ctx.kminmer_index)(d_reads, d_unitigs, k, dprev)
    rec, vec = t.to_host()
    assert np.array_equal(formats.sorted_abundance_records(rec.tobytes()), fx["abundance_sorted"])
    if fx["min_sorted"] is not None:
        assert np.array_equal(formats.sorted_vector_records(vec.astype("<u4").tobytes(), k), fx["min_sorted"])
    flags = ctx.small_contigs(d_unitigs, k, P.prev_k, dprev) if k > 8 else np.zeros(len(uo) - 1, np.uint8)
    mine = sorted((0, tuple(int(x) for x in um[int(uo[i]): int(uo[i + 1])])) for i in np.nonzero(flags)[0])
    assert mine == mk.small_contig_records(fx["small_contigs"])


@pytest.mark.parametrize("name", ["hifi_multik", "ont_multik"])
def test_edge_indexes_at_first_k_plus_one_vs_reference_log(ctx, name):
    """N2 once more at k = firstK+1, where the reference still has vectors: EdgeIndexer over the refined table and
    UnitigEdgeIndexer over the reference's own unitigGraph.nodes.bin against the counts and the checksum it logs."""
    from tests import multik_fixture as mk
    m = mk.manifest(name)
    k = m["first_k"] + 1
    log = m["per_k"][str(k)]["reference_log"]
    fx = mk.load(name, k)
    P = fx["params"]
    (rm, ro), (um, uo) = fx["reads"], fx["unitigs"]
    dprev = ctx.prev_from_records(fx["prev_records"])
    if fx["prev_unitigs"]:
        pm = np.concatenate([u for u, _ in fx["prev_unitigs"]]).astype(np.uint32)
        po = np.concatenate([[0], np.cumsum([len(u) for u, _ in fx["prev_unitigs"]])]).astype(np.uint64)
        ctx.prev_overlay_unitigs(dprev, ctx.minimizers_from_host(pm, po), np.array([a for _, a in fx["prev_unitigs"]], dtype=np.uint32), P.prev_k)
    table = ctx.kminmer_count_refined(ctx.minimizers_from_host(rm, ro), ctx.minimizers_from_host(um, uo), k, dprev)
    edges, ck = ctx.edge_index(table)
    assert edges.info()["n_records"] == log["n_edges"] and ck == log["edge_checksum"]
    nm, no, _ = formats.parse_unitig_nodes(open(os.path.join(fx["dir"], "unitigGraph.nodes.bin"), "rb").read())
    uedges, _ = ctx.unitig_edge_index(ctx.minimizers_from_host(nm, no), k)
    assert uedges.info()["n_records"] == log["n_unitig_edges"]


@pytest.mark.parametrize("mode", ["rccl", "peer", "auto"])
def test_library_exchange_one_rank(ctx, orc, mode):
    """The exchange inside the library (mdbg_comm_create_mode, mdbg_shard_exchange, mdbg_kminmer_count_first_sharded) over either
    transport -- RCCL send / receive groups on the context's stream, or peer copies with their shared control block, self-test and
    status phases -- with a communicator of one rank: the table must be the single-GPU table.  (More ranks: tests/test_gpu_multirank.py.)"""
    from metamdbg_amd import capi
    spec = synth.hifi_spec(3000, seed=17, read_len=6000, coverage=25.0)
    reads = ctx.reads_synthetic(spec)
    corr = ctx.purge_palindromes(ctx.scan(reads, K=15, density=0.005, hpc=True), 4, 100)
    rec0, vec0 = ctx.kminmer_count_first(corr, 4, 0).to_host()
    comm = ctx.comm_create(capi.Context.comm_unique_id(), 0, 1, mode)
    assert comm.mode == ("rccl" if mode == "rccl" else "peer") and comm.note == ""
    try:
        for min_ab in (0, 2):
            exp_rec, exp_vec = (rec0, vec0) if min_ab == 0 else ctx.kminmer_count_first(corr, 4, min_ab).to_host()
            rec, vec = ctx.kminmer_count_first_sharded(comm, corr, 4, min_ab).to_host()
            assert np.array_equal(formats.sorted_abundance_records(rec), formats.sorted_abundance_records(exp_rec))
            assert np.array_equal(formats.sorted_vector_records(vec.astype("<u4").tobytes(), 4), formats.sorted_vector_records(exp_vec.astype("<u4").tobytes(), 4))
        # the same in three calls, as a caller with several batches in flight drives it
        sh = ctx.shard_begin(corr, 4, 1)
        rec, vec = sh.finish(sh.exchange(comm), 0).to_host()
        sh.free()
        assert np.array_equal(formats.sorted_abundance_records(rec), formats.sorted_abundance_records(rec0))
        hc = corr.to_host(full=False)
        t = orc.kminmer_count_first(hc["minimizers"], hc["offsets"], 4, 0)
        assert np.array_equal(formats.sorted_abundance_records(rec), formats.sorted_abundance_records(orc.table_abundance_records(t)))
    finally:
        comm.destroy()


def test_hifi_1m_digests(ctx):
    """BASELINE.json configs[1] at its stated size -- 1 M synthetic HiFi reads x 10 kb, single k iteration -- against the reference
    itself: tests/golden/hifi_1m/manifest.json holds the digests of what the reference's readSelection + graph --firstpass wrote
    for this read set in the build container (tests/golden/make_golden.py --only-1m: sha256 of read_data_init.txt, order-
    independent digests of read_data_corrected.txt and of both tables, the counts it logged, the abundance checksum).  The reads
    are regenerated on the device from the seed (bit-identical to the FASTA the reference read: manifest fasta_sha256 is of the
    host generator's file, test_synthetic_generator_matches_numpy ties the two) and put through the HIP path."""
    import hashlib
    path = os.path.join(H.GOLDEN, "hifi_1m", "manifest.json")
    if not os.path.exists(path):
        pytest.skip("tests/golden/hifi_1m/manifest.json not generated")
    with open(path) as f:
        g = json.load(f)
    spec = synth.hifi_spec(g["n_reads"], seed=g["seed"], read_len=g["read_len"], coverage=50.0)
    assert spec.species_len == g["species_len"]
    reads = ctx.reads_synthetic(spec)
    mins = ctx.scan(reads, K=g["K"], density=g["density"], hpc=bool(g["hpc"]))
    reads.free()
    init = formats.build_read_data_init(mins.to_host())
    assert len(init) == g["read_data_init_bytes"]
    assert hashlib.sha256(init).hexdigest() == g["read_data_init_sha256"]
    del init
    corr = ctx.purge_palindromes(mins, 4, 100)
    hc = corr.to_host(full=False)
    assert len(hc["minimizers"]) == g["n_corrected_minimizers"]
    assert formats.minimizer_reads_digest(hc["minimizers"], hc["offsets"]) == g["read_data_corrected_digest"]
    table = ctx.kminmer_count_first(corr, g["k"], g["min_abundance"])
    info = table.info()
    assert info["n_solid"] == g["reference_log"]["n_solid"] and info["n_records"] - info["n_solid"] == g["reference_log"]["n_rescued"]
    assert info["n_records"] == g["n_records"]
    sums = table.checksum()
    assert sums[0] == g["abundance_checksum"] and sums[1] == g["sum_abundance"]
    rec, vec = table.to_host()
    d = formats.table_digests(rec, vec.astype("<u4").tobytes(), g["k"])
    assert d["abundance_sorted_sha256"] == g["abundance_sorted_sha256"] and d["min_sorted_sha256"] == g["min_sorted_sha256"]
    for o in (table, corr, mins):
        o.free()


def test_ont_1m_digests(ctx):
    """BASELINE.json configs[3]'s kind of input tied to the reference AT SIZE (round-5 VERDICT item 4): 1,000,100 synthetic ONT reads x 20 kb
    with qualities, ONE file, so that the repetitive-minimizer census meets its cap -- the first 1,000,001 reads of a file are counted
    (readSelection/ReadSelection.hpp:497-561; Commons.hpp:5873: `readIndexPerDataset > _maxReads`), the last 99 are scanned but not counted --
    against the digests of what the reference's own `readSelection --skip-correction` + `graph --firstpass` wrote for this read set in the
    build container (tests/golden/make_golden.py --only-ont-1m; tests/golden/ont_1m/manifest.json): the census's pick, sha256 of
    read_data_init.txt (1 GB), the corrected reads, both tables, the counts it logged.  The census's order among equal counts at the cut is
    the one thing the reference leaves open (std::sort): when the device's pick differs from the reference's, the difference must lie among
    values whose counts tie at the cut -- checked against an independent count -- and the scan then runs with the reference's pick."""
    import hashlib
    path = os.path.join(H.GOLDEN, "ont_1m", "manifest.json")
    if not os.path.exists(path):
        pytest.skip("tests/golden/ont_1m/manifest.json not generated")
    with open(path) as f:
        g = json.load(f)
    info = ctx.device_info()
    if info["hbm_bytes"] < 200e9:
        pytest.skip("needs an MI355X's HBM")
    ctx.set_option("pool_trim", 1)
    spec = synth.ont_spec(g["n_reads"], seed=g["seed"], read_len=g["read_len"], coverage=g["coverage"])
    assert spec.species_len == g["species_len"] and g["n_reads"] > 1_000_001
    reads = ctx.reads_synthetic(spec)
    # the census: the first 1,000,001 reads of the file, at the correction density, qualities not looked at (CountMinimizerFunctor, :565-625)
    counted = ctx.reads_synthetic(spec, first_read=0, n_reads=1_000_001)
    cm = ctx.scan(counted, K=g["K"], density=g["correction_density"], hpc=False, apply_read_filters=False, ignore_qualities=True)
    counted.free()
    picked = ctx.repetitive_minimizers(cm)
    rep_ref = np.array(g["repetitive_minimizers"], dtype=np.uint32)
    assert len(picked) == len(rep_ref)
    if set(picked.tolist()) != set(rep_ref.tolist()):
        vals = cm.to_host(full=False)["minimizers"]
        odd = np.array(sorted(set(picked.tolist()) ^ set(rep_ref.tolist())), dtype=np.uint32)
        both = np.array(sorted(set(picked.tolist()) | set(rep_ref.tolist())), dtype=np.uint32)
        at = np.searchsorted(both, vals)
        at[at == len(both)] = 0
        cnt = np.bincount(at[both[at] == vals], minlength=len(both))
        count_of = dict(zip(both.tolist(), cnt.tolist()))
        cut = min(count_of[int(v)] for v in rep_ref)
        assert all(count_of[int(v)] == cut for v in odd), ("the picks differ beyond a tie at the cut", [(int(v), count_of[int(v)]) for v in odd][:20], cut)
        del vals
    cm.free()
    # the pass itself, every read, with the reference's pick
    mins = ctx.scan(reads, K=g["K"], density=g["density"], hpc=False, repetitive=rep_ref)
    reads.free()
    init = formats.build_read_data_init(mins.to_host())
    assert len(init) == g["read_data_init_bytes"]
    assert hashlib.sha256(init).hexdigest() == g["read_data_init_sha256"]
    del init
    # --skip-correction: the purge runs over the scan's output (ReadSelection.hpp:300, :1374-1431)
    n50 = formats.parse_read_stats(bytes.fromhex(g["read_stats_hex"]))["n50"]
    last_k = max(int(np.float32(n50) * np.float32(g["density"]) * np.float32(2.0)), 6)          # Commons::computeLastK (Commons.hpp:1726-1741), no --max-k
    assert last_k == 200
    corr = ctx.purge_palindromes(mins, 4, last_k)
    hc = corr.to_host(full=False)
    assert len(hc["minimizers"]) == g["n_corrected_minimizers"]
    assert formats.minimizer_reads_digest(hc["minimizers"], hc["offsets"]) == g["read_data_corrected_digest"]
    del hc
    table = ctx.kminmer_count_first(corr, g["k"], g["min_abundance"])
    ti = table.info()
    assert ti["n_solid"] == g["reference_log"]["n_solid"] and ti["n_records"] - ti["n_solid"] == g["reference_log"]["n_rescued"]
    assert ti["n_records"] == g["n_records"]
    sums = table.checksum()
    assert sums[0] == g["abundance_checksum"] and sums[1] == g["sum_abundance"]
    rec, vec = table.to_host()
    d = formats.table_digests(rec, vec.astype("<u4").tobytes(), g["k"])
    assert d["abundance_sorted_sha256"] == g["abundance_sorted_sha256"] and d["min_sorted_sha256"] == g["min_sorted_sha256"]
    for o in (table, corr, mins):
        o.free()
    ctx.set_option("pool_trim", 1)


def test_minimizers_concat(ctx):
    """mdbg_minimizers_concat: a read set scanned in pieces (fresh scattered scan outputs, a piece without reads, qualities) and
    appended on the device is the set scanned whole -- every array of the scan output -- and so are its purge and its table;
    purged pieces (values and offsets only) append as well."""
    spec = synth.ont_spec(3000, seed=31, read_len=9000, coverage=30.0)
    cuts = [(0, 700), (700, 0), (700, 1800), (2500, 500)]
    whole = ctx.reads_synthetic(spec)
    kw = dict(K=15, density=0.005, hpc=False)
    mw = ctx.scan(whole, **kw)
    parts = [ctx.scan(ctx.reads_synthetic(spec, first_read=f, n_reads=n), **kw) for f, n in cuts]
    cat = ctx.minimizers_concat(parts)
    hw, hc = mw.to_host(), cat.to_host()
    assert sorted(hw) == sorted(hc)
    for key in hw:
        assert np.array_equal(hw[key], hc[key], equal_nan=True) if hw[key].dtype.kind == "f" else np.array_equal(hw[key], hc[key]), key
    assert len(hw["minimizers"]) > 100_000 and hw["qual"].max() > 1
    tw = ctx.kminmer_count_first(ctx.purge_palindromes(mw, 4, 100), 4, 0)
    tc = ctx.kminmer_count_first(ctx.purge_palindromes(cat, 4, 100), 4, 0)
    assert tw.checksum() == tc.checksum() and tw.info() == tc.info()
    # purged pieces: CSR with values only
    pc = ctx.minimizers_concat([ctx.purge_palindromes(p, 4, 100) for p in parts])
    tp = ctx.kminmer_count_first(pc, 4, 0)
    assert tp.checksum() == tw.checksum() and pc.info() == dict(n_reads=3000, n_minimizers=tw.stats()["minimizers"])
    assert ctx.minimizers_concat([]).info() == dict(n_reads=0, n_minimizers=0)


def test_bulk_export_of_bases_and_qualities(ctx):
    """mdbg_reads_export_ascii (four bases per look-up, lengths that are not multiples of four) and mdbg_reads_export_qualities
    against the per-read mdbg_reads_get, on a sub-range of a batch with qualities."""
    rng = np.random.default_rng(3)
    seqs = [bytes(synth.CODE2ASCII[rng.integers(0, 4, int(n))]) for n in (1, 2, 3, 4, 5, 31, 32, 33, 63, 64, 65, 1000, 1001, 1002, 1003, 0, 7)]
    quals = [bytes(rng.integers(33, 90, len(s)).astype(np.uint8)) for s in seqs]
    reads = ctx.reads_from_ascii(seqs, quals)
    for first, count in ((0, len(seqs)), (3, 9), (15, 2), (5, 0)):
        bases, offs = reads.export_ascii(first, count)
        q = reads.export_qualities(first, count)
        assert len(q) == len(bases) == sum(len(s) for s in seqs[first:first + count])
        for i in range(count):
            a, b = int(offs[i]), int(offs[i + 1])
            assert bases[a:b].tobytes() == seqs[first + i] and q[a:b].tobytes() == quals[first + i]
            assert reads.get(first + i, with_quality=True) == (seqs[first + i], quals[first + i])


def test_async_packed_upload_with_qualities(ctx):
    """mdbg_reads_from_packed_async + mdbg_reads_attach_qualities_async: the upload queued on the context's upload stream, the scan
    ordered after it on the device; several batches queued ahead of their scans; the result is the synchronous path's, and the
    length check of the qualities fails at once."""
    from metamdbg_amd import capi
    rng = np.random.default_rng(11)
    batches = []
    for _ in range(3):
        seqs = [synth.CODE2ASCII[rng.integers(0, 4, int(n))] for n in rng.integers(500, 12_000, 300)]
        quals = [rng.integers(33, 75, len(s)).astype(np.uint8) for s in seqs]
        words, woff, lens = synth.pack_reads([synth.ascii_to_codes(s) for s in seqs])
        qoff = np.zeros(len(seqs) + 1, dtype=np.uint64)
        np.cumsum([len(q) for q in quals], out=qoff[1:])
        batches.append((seqs, quals, np.ascontiguousarray(words, np.uint64), np.ascontiguousarray(woff, np.uint64),
                        np.ascontiguousarray(lens, np.uint32), b"".join(q.tobytes() for q in quals), qoff))
    kw = dict(K=15, density=0.01, hpc=True)
    queued = [ctx.reads_from_packed_async(b[2], b[3], b[4], b[5], b[6]) for b in batches]        # all three uploads in flight
    for b, r in zip(batches, queued):
        got = ctx.scan(r, **kw).to_host()
        want = ctx.scan(ctx.reads_from_ascii([s.tobytes() for s in b[0]], [q.tobytes() for q in b[1]]), **kw).to_host()
        for key in want:
            assert np.array_equal(got[key], want[key], equal_nan=True) if want[key].dtype.kind == "f" else np.array_equal(got[key], want[key]), key
        assert got["qual"].max() > 1 and len(got["minimizers"]) > 1000
        r.wait()
        assert r.get(7, with_quality=True) == (b[0][7].tobytes(), b[1][7].tobytes())
    b = batches[0]
    bad = b[6].copy(); bad[5] += 1
    with pytest.raises(capi.MdbgError):
        ctx.reads_from_packed_async(b[2], b[3], b[4], b[5], bad)


def test_few_marked_reads_are_routed_one_by_one(ctx, orc):
    """A batch in which a few reads carry an N, lower case or an IUPAC letter stays on the block-structured kernel; those reads alone
    are counted, placed and scanned by the general kernel (mdbg_scan, ScanArgs::skip) -- together with a read that outgrows the block
    kernel's stage, the other kind of read the general kernel finishes.  Against the oracle (reference semantics pinned by the
    scan_n / scan_case fixtures), with and without qualities, HPC on and off; once through mdbg_reads_from_ascii and once as a
    packed batch whose odd reads are handed again as characters (mdbg_reads_mark_ascii: what the host feed does)."""
    rng = np.random.default_rng(41)
    seqs = [bytearray(synth.CODE2ASCII[rng.integers(0, 4, int(n))]) for n in rng.integers(300, 7000, 500)]
    seqs[250] = bytearray(synth.CODE2ASCII[rng.integers(0, 4, 150_000)])        # selects more than the stage holds at this density
    odd = sorted(int(x) for x in rng.choice(500, 14, replace=False) if int(x) != 250)
    for j, r in enumerate(odd):
        sq = seqs[r]
        a = int(rng.integers(0, len(sq) - 40))
        if j % 3 == 0:
            sq[a:a + int(rng.integers(1, 6))] = b"N" * 5
        elif j % 3 == 1:
            sq[a:a + 30] = bytes(sq[a:a + 30]).lower()
        else:
            sq[a] = ord(rng.choice(list("RSWVBn")))
    odd.append(499); seqs[499][0] = ord("N")                                       # first base of the last read
    odd = sorted(set(odd))
    seqs = [bytes(x) for x in seqs]
    quals = [bytes(rng.integers(33, 80, len(x)).astype(np.uint8)) for x in seqs]
    for hpc in (True, False):
        for q in (None, quals):
            m, h = _check_scan_against_oracle(ctx, orc, seqs, q, 15, 0.005, hpc)
            assert int(np.diff(h["offsets"])[250]) > 384                       # the long read did outgrow the stage
            # the same batch packed on the host (code (c >> 1) & 3 for any character), the odd reads marked afterwards
            words, woff, lens = synth.pack_reads([synth.ascii_to_codes(x) for x in seqs])
            reads = ctx.reads_from_packed(words, woff, lens, [qq for qq in q] if q else None)
            reads.mark_ascii(odd, [seqs[r] for r in odd])
            h2 = ctx.scan(reads, K=15, density=0.005, hpc=hpc).to_host()
            assert formats.build_read_data_init(h2) == formats.build_read_data_init(h)
    # marking a read that turns out plain is harmless; indices must ascend and lengths must fit
    from metamdbg_amd import capi
    words, woff, lens = synth.pack_reads([synth.ascii_to_codes(x) for x in seqs[:20]])
    reads = ctx.reads_from_packed(words, woff, lens)
    reads.mark_ascii([3, 7], [seqs[3].upper().replace(b"N", b"A").replace(b"R", b"A"), seqs[7]])
    with pytest.raises(capi.MdbgError):
        ctx.reads_from_packed(words, woff, lens).mark_ascii([7, 3], [seqs[7], seqs[3]])
    with pytest.raises(capi.MdbgError):
        ctx.reads_from_packed(words, woff, lens).mark_ascii([3], [seqs[3][:-1]])


def test_context_options_do_not_change_results(ctx):
    """The tuning options of mdbg_set_option change how the work is laid out on the device, never what comes out: the table kernels
    as a handful of workgroups (table_grid_blocks) or one block per CU, every kernel but the block-structured scan confined to 16 CUs
    with the scan on a stream of its own (table_cu_count), the memory pool trimmed or allowed 90 % of the device, the scan capped at four or
    three blocks per CU by unused LDS (scan_lds_pad), the first pass on one table or partitioned in either of its kernel forms."""
    from metamdbg_amd import capi
    spec = synth.hifi_spec(4000, seed=61, read_len=8000, coverage=30.0)
    base = capi.Context(0)
    reads = base.reads_synthetic(spec)
    want_m = base.scan(reads, K=15, density=0.005, hpc=True).to_host()
    corr = base.purge_palindromes(base.scan(reads, K=15, density=0.005, hpc=True), 4, 100)
    want = (base.kminmer_count_first(corr, 4, 0).checksum(), base.kminmer_index(corr, None, 6, base.kminmer_count_refined(corr, None, 5, base.kminmer_count_first(corr, 4, 0))).checksum())
    for options in ({"table_grid_blocks": 7}, {"table_blocks_per_cu": 1}, {"table_cu_count": 16}, {"table_cu_count": 16, "table_grid_blocks": 64},
                    {"pool_cache_percent": 90, "pool_trim": 1}, {"scan_reads_per_wave": 5},
                    # round 4: what a context sharing its device with another batch's scan is given (bench.py), and the first pass's two paths
                    {"scan_lds_pad": 3072, "partition_tile": 2048, "partition_slot_list": 0}, {"scan_lds_pad": 10304}, {"scan_lds_reserve": 28672, "partition_lds_slots": 1024},
                    {"first_pass_mode": 1}, {"first_pass_mode": 2}, {"first_pass_mode": 2, "partition_tile": 2048, "partition_slot_list": 0},
                    {"first_pass_mode": 2, "partition_lds_slots": 2048, "partition_slot_list": 0}, {"partition_auto_min": 1}):
        c = capi.Context(0)
        for name, value in options.items():
            c.set_option(name, value)
        r = c.reads_synthetic(spec)
        m = c.scan(r, K=15, density=0.005, hpc=True)
        h = m.to_host()
        for key in want_m:
            assert np.array_equal(h[key], want_m[key], equal_nan=True) if want_m[key].dtype.kind == "f" else np.array_equal(h[key], want_m[key]), (options, key)
        cr = c.purge_palindromes(m, 4, 100)
        t4 = c.kminmer_count_first(cr, 4, 0)
        got = (t4.checksum(), c.kminmer_index(cr, None, 6, c.kminmer_count_refined(cr, None, 5, t4)).checksum())
        assert got == want, options
        if "table_cu_count" in options:                 # and back: one unconfined stream again
            c.set_option("table_cu_count", 0)
            assert c.kminmer_count_first(cr, 4, 0).checksum() == want[0]
        c.close()
    base.close()


def test_table_checksum_is_the_references_formula(ctx):
    """mdbg_table_checksum on the device = the sums over the host copy of the rows; sums[0] is the "Checksum kminmer abundance" the
    reference logs when it loads a table (graph/CreateMdbg.cpp:3321: abundance * vecHash truncated to u64 -- the low word)."""
    spec = synth.hifi_spec(2500, seed=5, read_len=7000, coverage=25.0)
    corr = ctx.purge_palindromes(ctx.scan(ctx.reads_synthetic(spec), K=15, density=0.005, hpc=True), 4, 100)
    for table in (ctx.kminmer_count_first(corr, 4, 0), ctx.kminmer_count_first(corr, 5, 2)):
        rec, vec = table.to_host()
        lo, hi, ab = rec["lo"].astype(np.uint64), rec["hi"].astype(np.uint64), rec["abundance"].astype(np.uint64)
        with np.errstate(over="ignore"):
            w = (vec.astype(np.uint64) * (2 * np.arange(vec.shape[1], dtype=np.uint64) + 1)).sum(axis=1, dtype=np.uint64)
            exp = (int((ab * lo).sum(dtype=np.uint64)), int(ab.sum(dtype=np.uint64)), int(hi.sum(dtype=np.uint64)),
                   int((w * (lo | np.uint64(1))).sum(dtype=np.uint64)))
        assert table.checksum() == exp and len(rec) > 1000


@pytest.mark.parametrize("mode", ["rccl", "peer"])
def test_exchange_failures_are_reported_and_leave_the_communicator_usable(ctx, mode):
    """A rank that fails locally inside mdbg_shard_exchange -- before the counts travel, allocating the receive buffers, in the
    owner's reduction (test_exchange_fail_phase 1 / 2 / 3) -- still takes part in the small agreement collectives, returns its own
    error, and leaves no RCCL group open: the next exchange on the same communicator works and gives the single-GPU table.
    mdbg_shard_abort (the caller's local half failed) behaves like phase 1.  With peers, they return MDBG_EPEER
    (tests/test_distributed_gloo.py runs the same protocol with two ranks).  Also: mdbg_comm_stats counts what travelled, and a
    corrupted reply (test_corrupt_replies) changes the table's checksum -- what the job-level self-checks look at."""
    from metamdbg_amd import capi
    spec = synth.hifi_spec(3000, seed=18, read_len=6000, coverage=25.0)
    corr = ctx.purge_palindromes(ctx.scan(ctx.reads_synthetic(spec), K=15, density=0.005, hpc=True), 4, 100)
    want = ctx.kminmer_count_first(corr, 4, 0)
    want_sum, want_info = want.checksum(), want.info()
    comm = ctx.comm_create(capi.Context.comm_unique_id(), 0, 1, mode)
    try:
        st0 = comm.stats()
        assert (st0["rank"], st0["n_ranks"], st0["rccl_ranks"], st0["exchanges"], st0["mode"]) == (0, 1, 1 if mode == "rccl" else 0, 0, mode)
        for phase, code in ((1, -1), (2, -3), (3, -4)):
            ctx.set_option("test_exchange_fail_phase", phase)
            sh = ctx.shard_begin(corr, 4, 1)
            with pytest.raises(capi.MdbgError) as ei:
                sh.exchange(comm)
            assert ei.value.code == code and "test failure" in str(ei.value)
            sh.free()
            # the communicator is as good as new
            sh = ctx.shard_begin(corr, 4, 1)
            t = sh.finish(sh.exchange(comm), 0)
            assert t.checksum() == want_sum and t.info() == want_info
            sh.free(); t.free()
        comm.abort(ctx, -3)                                  # no shard at all: the local half failed
        t = ctx.kminmer_count_first_sharded(comm, corr, 4, 0)
        assert t.checksum() == want_sum
        t.free()
        st = comm.stats()
        assert st["exchanges"] == 4 and st["bytes_to_peers"] == 0 and st["bytes_local"] > 0 and st["exchange_ms"] > 0
        # one reply off by one: the table this rank builds from it is not the single-GPU table any more
        ctx.set_option("test_corrupt_replies", 1)
        sh = ctx.shard_begin(corr, 4, 1)
        t = sh.finish(sh.exchange(comm), 0)
        assert t.checksum() != want_sum
        sh.free(); t.free()
    finally:
        comm.destroy()


def _emulate_exchange(sh_list):
    """Both all-to-alls of the sharded passes on one device (host copies stand in for the wire): rows to their owners,
    mdbg_shard_reduce on every owner, replies back to the senders in the order sent.  Returns one device buffer of replies
    per rank (the caller frees with hipFree) and the hip handle."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so.7")
    n_ranks = len(sh_list)
    rw = sh_list[0].row_words

    def to_host(ptr, shape):
        out = np.zeros(shape, dtype=np.uint64)
        if out.nbytes:
            assert hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(out.nbytes), 2) == 0
        return out

    def to_device(a):
        buf = C.c_void_p()
        assert hip.hipMalloc(C.byref(buf), C.c_size_t(max(a.nbytes, 8))) == 0
        if a.nbytes:
            assert hip.hipMemcpy(buf, a.ctypes.data_as(C.c_void_p), C.c_size_t(a.nbytes), 1) == 0
        return buf

    sent = [to_host(s.d_rows, (s.n_rows, rw)) for s in sh_list]
    per_owner = [[] for _ in range(n_ranks)]
    for r in range(n_ranks):
        o = 0
        for dst in range(n_ranks):
            c = int(sh_list[r].counts[dst])
            per_owner[dst].append(sent[r][o: o + c]); o += c
    replies = []
    for dst in range(n_ranks):
        rows = np.ascontiguousarray(np.concatenate(per_owner[dst]))
        buf = to_device(rows)
        replies.append(to_host(sh_list[dst].reduce(buf.value, len(rows)), (len(rows),)))
        hip.hipFree(buf)
    out = []
    for r in range(n_ranks):
        glob = []
        for dst in range(n_ranks):
            o = sum(int(sh_list[src].counts[dst]) for src in range(r))
            glob.append(replies[dst][o: o + int(sh_list[r].counts[dst])])
        out.append(to_device(np.ascontiguousarray(np.concatenate(glob))))
    return out, hip


@pytest.mark.parametrize("n_ranks", [2, 3])
def test_sharded_next_k_on_one_gpu(ctx, orc, n_ranks):
    """BASELINE.json configs[2] on more than one GPU: k = firstK+1 (refined) and the index passes with the reads sharded.  Every
    rank holds the whole previous table and runs the ordinary pass over ITS reads; mdbg_shard_from_table -> reduce -> mdbg_shard_keep
    make the ranks agree on who lists a key several of them found (exchanges emulated on one device).  The union of the kept
    tables must be the single-GPU table at every k, vectors included at firstK+1, and it is the next k's previous table."""
    rng = np.random.default_rng(31 + n_ranks)
    genome = rng.permutation(2500).astype(np.uint32)          # reads = windows of one minimizer-space genome, both strands, ~12x
    rl = []
    for _ in range(700):
        a = int(rng.integers(0, 2450)); n = int(rng.integers(0, 70))
        seg = genome[a:a + n]
        rl.append(seg[::-1].copy() if rng.integers(0, 2) else seg.copy())
    offs = np.concatenate([[0], np.cumsum([len(x) for x in rl])]).astype(np.uint64)
    mins = np.concatenate(rl).astype(np.uint32)
    n_reads = len(offs) - 1
    cuts = np.linspace(0, n_reads, n_ranks + 1).astype(int)
    shards = [ctx.minimizers_from_host(mins[int(offs[cuts[r]]): int(offs[cuts[r + 1]])], offs[cuts[r]: cuts[r + 1] + 1] - offs[cuts[r]])
              for r in range(n_ranks)]
    whole = ctx.minimizers_from_host(mins, offs)
    prev_rec, _ = ctx.kminmer_count_first(whole, 4, 0).to_host()
    for k in (5, 6, 7):
        prev = ctx.prev_from_records(prev_rec)
        single = ctx.kminmer_count_refined(whole, None, k, prev) if k == 5 else ctx.kminmer_index(whole, None, k, prev)
        exp_rec, exp_vec = single.to_host()
        assert len(exp_rec) > 50
        locals_ = [ctx.kminmer_count_refined(s, None, k, prev) if k == 5 else ctx.kminmer_index(s, None, k, prev) for s in shards]
        assert sum(t.info()["n_records"] for t in locals_) > len(exp_rec)          # the shards do overlap in keys
        sh = [ctx.shard_from_table(t, n_ranks) for t in locals_]
        bufs, hip = _emulate_exchange(sh)
        recs, vecs = [], []
        for r in range(n_ranks):
            kept = sh[r].keep(bufs[r].value)
            rec, vec = kept.to_host()
            recs.append(rec); vecs.append(vec)
            hip.hipFree(bufs[r]); sh[r].free(); kept.free()
        rec = np.concatenate(recs)
        assert np.array_equal(formats.sorted_abundance_records(rec), formats.sorted_abundance_records(exp_rec)), k
        if exp_vec is not None:
            v = np.concatenate(vecs)
            assert np.array_equal(formats.sorted_vector_records(v.astype("<u4").tobytes(), k), formats.sorted_vector_records(exp_vec.astype("<u4").tobytes(), k)), k
        # against the oracle too
        oprev = orc.PrevAbundance(prev_rec.tobytes()); oprev.overlay_unitigs([], k - 1)
        t_orc = (orc.kminmer_count_refined if k == 5 else orc.kminmer_index)(mins, offs, k, oprev)
        assert np.array_equal(formats.sorted_abundance_records(rec), formats.sorted_abundance_records(orc.table_abundance_records(t_orc))), k
        prev_rec = rec                         # what an all-gather of the ranks' records gives every rank


def test_full_size_properties_ont(ctx, orc):
    """BASELINE.json configs[3] preset at a FULL-size batch (1 M x 20 kb ONT reads with qualities = 20 Gbp: 1 % substitutions,
    0.5 % insertions, 0.5 % deletions; no HPC, repetitive filter from the 0.025 census), checked through properties:
      * the census (over the first 200 000 reads) picks max(1, floor(1e-5 x distinct)) values, each with a count at or above
        the cut of an independent count;
      * the oracle on a sample of the batch equals the corresponding slice of the full scan (values, positions, directions,
        per-minimizer minimum qualities, mean read quality, lengths);
      * shard invariance of the scan (the two halves scanned separately, with the same repetitive set); purge idempotent;
      * on the first 200 000 reads (the table of 1 M such reads holds 75 M records: comparing them as sorted multisets on the
        host would take minutes): linearity of the counts -- the sharded first pass over two halves (exchanges emulated)
        equals the single call as a multiset; rescued rows carry abundance 1 and dominate (2 % errors)."""
    n, L, k, n_sub = 1_000_000, 20_000, 4, 200_000
    spec = synth.ont_spec(n, seed=44, read_len=L, coverage=50.0)
    head = ctx.reads_synthetic(spec, first_read=0, n_reads=n_sub)
    pre = ctx.scan(head, K=15, density=0.025, hpc=False, apply_read_filters=False)
    rep = ctx.repetitive_minimizers(pre)
    vals, counts = np.unique(pre.to_host(full=False)["minimizers"], return_counts=True)
    pre.free(); head.free()
    n_keep = max(int(np.float32(0.00001) * np.float32(len(vals))), 1)
    cut = np.sort(counts)[::-1][n_keep - 1]
    top = set(vals[counts >= cut].tolist())
    assert len(rep) == n_keep > 50 and len(set(rep.tolist())) == n_keep and all(int(v) in top for v in rep)
    assert set(vals[counts > cut].tolist()) <= set(rep.tolist())
    del vals, counts
    reads = ctx.reads_synthetic(spec)
    full = ctx.scan(reads, K=15, density=0.005, hpc=False, repetitive=rep)
    h = full.to_host()
    assert int(h["offsets"][-1]) == len(h["minimizers"]) > 90 * n
    for first in (0, n - 200):
        for r in range(0, 200, 9):
            b, q = reads.get(first + r, with_quality=True)
            o = orc.read_selection(b, q, K=15, density=0.005, hpc=False, repetitive=rep)
            a, e = int(h["offsets"][first + r]), int(h["offsets"][first + r + 1])
            assert h["minimizers"][a:e].tolist() == o["minimizers"].tolist() and h["pos"][a:e].tolist() == o["pos"].tolist()
            assert h["dir"][a:e].tolist() == o["dir"].tolist() and h["qual"][a:e].tolist() == o["qual"].tolist()
            assert _nan_eq([h["mean_quality"][first + r]], [o["mean_quality"]]) and int(h["read_length"][first + r]) == L
    reads.free()
    cutm = int(h["offsets"][n // 2])
    for f, lo, hi in ((0, 0, cutm), (n // 2, cutm, len(h["minimizers"]))):
        part = ctx.reads_synthetic(spec, first_read=f, n_reads=n // 2)
        m = ctx.scan(part, K=15, density=0.005, hpc=False, repetitive=rep)
        part.free()
        hp = m.to_host(full=False)
        assert np.array_equal(hp["minimizers"], h["minimizers"][lo:hi])
        assert np.array_equal(hp["offsets"], h["offsets"][f: f + n // 2 + 1] - h["offsets"][f])
        m.free()
    corr = ctx.purge_palindromes(full, 4, 200)
    corr2 = ctx.purge_palindromes(corr, 4, 200)
    hc, hc2 = corr.to_host(full=False), corr2.to_host(full=False)
    assert np.array_equal(hc["minimizers"], hc2["minimizers"]) and np.array_equal(hc["offsets"], hc2["offsets"])
    corr2.free(); full.free(); corr.free()
    # tables: the first n_sub reads, whole and as two shards
    o_sub = hc["offsets"][: n_sub + 1]
    m_sub = hc["minimizers"][: int(o_sub[-1])]
    whole = ctx.minimizers_from_host(m_sub, o_sub)
    t = ctx.kminmer_count_first(whole, k, 0)
    rec, vec = t.to_host()
    info = t.info()
    assert (rec[: info["n_solid"]]["abundance"] > 1).all() and (rec[info["n_solid"]:]["abundance"] == 1).all()
    assert info["n_records"] - info["n_solid"] > info["n_solid"]
    t.free(); whole.free()
    mid = n_sub // 2
    shards = [ctx.minimizers_from_host(m_sub[: int(o_sub[mid])], o_sub[: mid + 1]),
              ctx.minimizers_from_host(m_sub[int(o_sub[mid]):], o_sub[mid:] - o_sub[mid])]
    sh = [ctx.shard_begin(s_, k, 2) for s_ in shards]
    bufs, hip = _emulate_exchange(sh)
    recs, vecs, n_solid = [], [], 0
    for r in range(2):
        ts = sh[r].finish(bufs[r].value, 0)
        hip.hipFree(bufs[r])
        rs, vs = ts.to_host()
        recs.append(rs); vecs.append(vs); n_solid += ts.info()["n_solid"]
        ts.free(); sh[r].free()
    assert n_solid == info["n_solid"]
    assert np.array_equal(formats.sorted_abundance_records(np.concatenate(recs)), formats.sorted_abundance_records(rec))
    assert np.array_equal(formats.sorted_vector_records(np.concatenate(vecs).astype("<u4").tobytes(), k),
                          formats.sorted_vector_records(vec.astype("<u4").tobytes(), k))


@pytest.mark.parametrize("hpc,with_q", [(True, False), (False, False), (False, True), (True, True)])
def test_scan_long_reads_among_short(ctx, orc, hpc, with_q):
    """A batch of ordinary reads with a few that select more minimizers than the block kernel's LDS stage holds (several
    hundred kb: ONT's long tail): those reads are placed behind the regions and re-run by the general kernel; every record,
    and the order of the reads, must be the oracle's."""
    rng = np.random.default_rng(5 + 2 * int(hpc) + int(with_q))
    lens = [int(x) for x in rng.integers(2000, 9000, 300)]
    for at, n in ((7, 210_000), (150, 400_000), (151, 120_000), (299, 95_000)):
        lens[at] = n
    seqs = [bytes(synth.CODE2ASCII[rng.integers(0, 4, n)]) for n in lens]
    quals = [bytes((rng.integers(2, 60, n) + 33).astype(np.uint8)) for n in lens] if with_q else None
    reads = ctx.reads_from_ascii(seqs, quals)
    m = ctx.scan(reads, K=15, density=0.005, hpc=hpc)
    h = m.to_host()
    counts = np.diff(h["offsets"].astype(np.int64))
    assert counts[150] > 1200 and counts[7] > 600          # really beyond the stage (384)
    exp = b"".join(orc.read_selection(s, quals[i] if with_q else None, K=15, density=0.005, hpc=hpc)["record"] for i, s in enumerate(seqs))
    assert formats.build_read_data_init(h) == exp
    # and through the purge, which reads the scattered form
    m2 = ctx.scan(reads, K=15, density=0.005, hpc=hpc)
    hc = ctx.purge_palindromes(m2, 4, 100).to_host(full=False)
    assert int(hc["offsets"][-1]) <= int(h["offsets"][-1]) and len(hc["offsets"]) == len(seqs) + 1
    want = [orc.purge_palindrome(h["minimizers"][int(h["offsets"][i]): int(h["offsets"][i + 1])], 4, 100) for i in (7, 150, 151, 0, 299)]
    for i, w in zip((7, 150, 151, 0, 299), want):
        assert hc["minimizers"][int(hc["offsets"][i]): int(hc["offsets"][i + 1])].tolist() == w.tolist()


@pytest.mark.parametrize("hpc,with_q", [(True, False), (False, False), (True, True)])
def test_scan_false_candidates_are_rerun(ctx, orc, hpc, with_q):
    """The block kernel records candidate positions by the upper half of the hash and confirms each with the full hash when it
    is materialised; a read with a false candidate (one position in 2^31) is re-run by the general kernel.  With the
    candidate test widened on purpose a few per cent of the reads take that way; the records must not change."""
    rng = np.random.default_rng(31 + 2 * int(hpc) + int(with_q))
    lens = [int(x) for x in rng.integers(3000, 9000, 3000)]
    if not hpc: lens[11] = 300_000                          # outgrows the stage as well: its count must still be exact
    seqs = [bytes(synth.CODE2ASCII[rng.integers(0, 4, n)]) for n in lens]
    quals = [bytes((rng.integers(2, 60, n) + 33).astype(np.uint8)) for n in lens] if with_q else None
    reads = ctx.reads_from_ascii(seqs, quals)
    plain = formats.build_read_data_init(ctx.scan(reads, K=15, density=0.005, hpc=hpc).to_host())
    ctx.set_option("scan_candidate_slack", 1 << 15)
    try:
        ctx.timing(True); ctx.timing_reset()
        widened = formats.build_read_data_init(ctx.scan(reads, K=15, density=0.005, hpc=hpc).to_host())
        launches = ctx.timing_get("scan")[1]
    finally:
        ctx.set_option("scan_candidate_slack", 0)
        ctx.timing(False)
    assert launches == 2                                    # the block kernel, then the general kernel over the reads it lost
    assert widened == plain
    exp = b"".join(orc.read_selection(s, quals[i] if with_q else None, K=15, density=0.005, hpc=hpc)["record"] for i, s in enumerate(seqs[:200]))
    assert plain[:len(exp)] == exp


def test_scan_ignoring_the_qualities_of_a_read_set(ctx):
    """ignore_qualities (the census of minimizer values over reads that are resident with their qualities,
    ReadSelection.hpp:565-625): the records are those of the same reads without qualities."""
    rng = np.random.default_rng(77)
    lens = [int(x) for x in rng.integers(50, 30000, 400)]
    seqs = [bytes(synth.CODE2ASCII[rng.integers(0, 4, n)]) for n in lens]
    quals = [bytes((rng.integers(2, 60, n) + 33).astype(np.uint8)) for n in lens]
    with_q, without = ctx.reads_from_ascii(seqs, quals), ctx.reads_from_ascii(seqs, None)
    for density in (0.005, 0.025):
        a = formats.build_read_data_init(ctx.scan(with_q, K=15, density=density, hpc=False, apply_read_filters=False, ignore_qualities=True).to_host())
        b = formats.build_read_data_init(ctx.scan(without, K=15, density=density, hpc=False, apply_read_filters=False).to_host())
        c = formats.build_read_data_init(ctx.scan(with_q, K=15, density=density, hpc=False, apply_read_filters=False).to_host())
        assert a == b and a != c


def test_scan_fastq_hpc_long_homopolymers(ctx, orc):
    """Qualities under HPC: the block kernel looks the original coordinates of a selected position up in the offsets of the
    last few tiles.  Homopolymer stretches of tens of kb squeeze many tiles into a few compressed positions, so listed
    positions lose their tile before they are materialised: those reads go to the general kernel.  Records as the oracle's."""
    rng = np.random.default_rng(123)
    seqs = []
    for i in range(60):
        parts = []
        for _ in range(int(rng.integers(1, 5))):
            parts.append(synth.CODE2ASCII[rng.integers(0, 4, int(rng.integers(500, 6000)))])
            if i % 3 != 2:       # every third read is ordinary
                parts.append(np.full(int(rng.integers(3000, 60000)), ord("ACGT"[int(rng.integers(0, 4))]), dtype=np.uint8))
        parts.append(synth.CODE2ASCII[rng.integers(0, 4, int(rng.integers(500, 6000)))])
        seqs.append(bytes(np.concatenate(parts)))
    quals = [bytes((rng.integers(2, 60, len(s)) + 33).astype(np.uint8)) for s in seqs]
    reads = ctx.reads_from_ascii(seqs, quals)
    # without the read filters (these reads are low-complexity by construction): the correction scan's form of the call,
    # ReadCorrection.hpp:2269-2372, with homopolymer compression
    h = ctx.scan(reads, K=15, density=0.005, hpc=True, apply_read_filters=False, quality_window=1).to_host()
    n_total = 0
    for i, s_ in enumerate(seqs):
        exp = orc.correction_scan(s_, quals[i], K=15, density=0.005, hpc=True)
        a, b = int(h["offsets"][i]), int(h["offsets"][i + 1])
        assert h["minimizers"][a:b].tolist() == exp["minimizers"].tolist(), i
        assert h["pos"][a:b].tolist() == exp["pos"].tolist(), i
        assert h["dir"][a:b].tolist() == exp["dir"].tolist(), i
        assert h["qual"][a:b].tolist() == exp["qual"].tolist(), i
        n_total += b - a
    assert n_total > 500


def test_full_size_fastq_hpc(ctx, orc):
    """The bench batch's size with qualities under HPC (1 M x 10 kb reads, FASTQ form): selection must not depend on the
    qualities (values, positions, directions equal the FASTA scan of the same bases), the per-minimizer minimum qualities and the
    mean read quality equal the oracle's on a sample, both quality windows, and no read may have been lost on the way (every
    read of such a batch stays in the block kernel: one launch)."""
    import dataclasses
    n, L = 1_000_000, 10_000
    spec = synth.hifi_spec(n, seed=42, read_len=L, coverage=50.0)
    plain = ctx.reads_synthetic(spec)
    hp = ctx.scan(plain, K=15, density=0.005, hpc=True).to_host()
    plain.free()
    reads = ctx.reads_synthetic(dataclasses.replace(spec, with_quality=True))
    for window in (0, 1):
        ctx.timing(True); ctx.timing_reset()
        hq = ctx.scan(reads, K=15, density=0.005, hpc=True, quality_window=window).to_host()
        launches = ctx.timing_get("scan")[1]
        ctx.timing(False)
        assert launches == 1
        for f in ("offsets", "minimizers", "pos", "dir", "read_length"):
            assert np.array_equal(hq[f], hp[f]), f
        for r in list(range(0, 2000, 41)) + list(range(n - 2000, n, 41)):
            b, q = reads.get(r, with_quality=True)
            o = orc.correction_scan(b, q, K=15, density=0.005, hpc=True) if window else orc.read_selection(b, q, K=15, density=0.005, hpc=True)
            a, e = int(hq["offsets"][r]), int(hq["offsets"][r + 1])
            assert hq["minimizers"][a:e].tolist() == o["minimizers"].tolist()
            assert hq["qual"][a:e].tolist() == o["qual"].tolist(), (window, r)
            if not window:
                assert _nan_eq([hq["mean_quality"][r]], [o["mean_quality"]])
    assert (hp["qual"] == 1).all()          # no qualities: ReadSelection.hpp:1047-1051


def test_library_exchange_one_rank_large_share(ctx):
    """A rank's own share of the rows beyond half a GB (one rank, a read set with hardly any repeated k-min-mer: some 25 M rows
    of 24 bytes).  RCCL's send / receive to self returned with 531 MiB of such a message in place and the rest unwritten; the
    library copies a rank's own share itself.  The sharded table must equal the plain one (order-independent digests)."""
    import dataclasses
    from metamdbg_amd import capi
    # 2 % errors: nine k-min-mers in ten are seen once, whatever the coverage
    spec = dataclasses.replace(synth.ont_spec(380_000, seed=23, read_len=20_000, coverage=50.0), with_quality=False)
    reads = ctx.reads_synthetic(spec)
    corr = ctx.purge_palindromes(ctx.scan(reads, K=15, density=0.005, hpc=False), 4, 100)
    reads.free()

    def digest(table):
        rec, _ = table.to_host()
        info = table.info()
        table.free()
        lo, hi, ab = rec["lo"].astype(np.uint64), rec["hi"].astype(np.uint64), rec["abundance"].astype(np.uint64)
        return (info["n_records"], info["n_solid"], int(np.bitwise_xor.reduce(lo)), int(np.bitwise_xor.reduce(hi)),
                int((lo * (ab + np.uint64(1))).sum(dtype=np.uint64)), int(ab.sum()))
    want = digest(ctx.kminmer_count_first(corr, 4, 0))
    comm = ctx.comm_create(capi.Context.comm_unique_id(), 0, 1)
    try:
        sh = ctx.shard_begin(corr, 4, 1)
        assert sh.n_rows * sh.row_words * 8 > 560 << 20          # the share really is beyond what arrived
        got = digest(sh.finish(sh.exchange(comm), 0))
        sh.free()
    finally:
        comm.destroy()
    assert got == want and want[0] > 20_000_000


def test_census_batch_by_batch(ctx):
    """mdbg_census_*: the census fed in batches (the table grows and is refilled on the way) picks what the one-call form
    picks over all the minimizers, and what an independent count says."""
    spec = synth.ont_spec(6000, seed=9, read_len=20_000, coverage=40.0)
    parts = [ctx.reads_synthetic(spec, first_read=f, n_reads=n) for f, n in ((0, 500), (500, 2500), (3000, 3000))]
    whole = ctx.reads_synthetic(spec)
    c = ctx.census()
    for r in parts:
        m = ctx.scan(r, K=15, density=0.025, hpc=False, apply_read_filters=False, ignore_qualities=True)
        c.add(m)
        m.free(); r.free()
    picked = c.top()
    c.free()
    mw = ctx.scan(whole, K=15, density=0.025, hpc=False, apply_read_filters=False, ignore_qualities=True)
    one_call = ctx.repetitive_minimizers(mw)
    assert picked.tolist() == one_call.tolist()
    vals, counts = np.unique(mw.to_host(full=False)["minimizers"], return_counts=True)
    n_keep = max(int(np.float32(0.00001) * np.float32(len(vals))), 1)
    order = np.lexsort((vals, -counts.astype(np.int64)))[:n_keep]
    exp = vals[order].tolist()
    cnt = dict(zip(vals.tolist(), counts.tolist()))
    assert n_keep >= 5 and len(picked) == n_keep, (n_keep, len(picked), len(vals))
    assert picked.tolist() == exp, [(int(v), cnt.get(int(v))) for v in picked][:20] + ["expected"] + [(v, cnt[v]) for v in exp][:20]


def test_census_of_any_values_and_growth(ctx):
    """The census counts in an array indexed by the value itself (4 bytes x the power of two above the largest value seen): values of
    any size up to 2^32 - 1 (l = 16), a later batch with larger values than the array was made for (it grows, the counts so far keep
    their places), ties at the cut by the smaller value, counts beyond the histogram's last bin -- against numpy."""
    rng = np.random.default_rng(77)
    def batch(hi, n, heavy):
        v = rng.integers(0, hi, n, dtype=np.uint64).astype(np.uint32)
        for val, times in heavy:
            v = np.concatenate([v, np.full(times, val, np.uint32)])
        rng.shuffle(v)
        cut = sorted(rng.choice(np.arange(1, len(v)), 40, replace=False).tolist())
        offs = np.array([0] + cut + [len(v)], np.uint64)
        return v, offs
    # first small values only, then a batch reaching 2^30, then one reaching 2^32 - 1; heavy values of 5000+ occurrences (> CENSUS_BINS)
    # and a tie at the cut: 150 001 distinct values or so -> keep = 1 .. 2
    b1 = batch(1 << 12, 50_000, [(7, 6000), (9, 6000)])
    b2 = batch(1 << 30, 400_000, [((1 << 30) - 1, 5000), (123_456_789, 7000)])
    b3 = batch((1 << 32) - 1, 400_000, [(0xFFFFFFFF, 6500), (0xFFFFFFFE, 100)])
    for upto in (1, 2, 3):
        bs = (b1, b2, b3)[:upto]
        c = ctx.census()
        for v, offs in bs:
            m = ctx.minimizers_from_host(v, offs)
            c.add(m)
            m.free()
        picked = c.top()
        c.free()
        allv = np.concatenate([v for v, _ in bs])
        vals, counts = np.unique(allv, return_counts=True)
        n_keep = max(int(np.float32(0.00001) * np.float32(len(vals))), 1)
        order = np.lexsort((vals, -counts.astype(np.int64)))[:n_keep]
        assert picked.tolist() == vals[order].tolist(), (upto, n_keep, picked.tolist(), vals[order].tolist())
    # every value distinct but two: the cut falls among the ties of count 1, the smaller values win
    v = np.arange(1000, 301_000, dtype=np.uint32); v = np.concatenate([v, v[-2:]])
    m = ctx.minimizers_from_host(v, np.array([0, len(v)], np.uint64))
    picked = ctx.repetitive_minimizers(m)
    m.free()
    n_keep = max(int(np.float32(0.00001) * np.float32(300_000)), 1)
    assert n_keep == 3 and picked.tolist() == [300_998, 300_999, 1000], (n_keep, picked.tolist())


def test_table_rows_in_pieces(ctx):
    """mdbg_table_to_host_range: any cut of the rows gives the rows of mdbg_table_to_host (the tool streams large tables through it)."""
    spec = synth.hifi_spec(1500, seed=8, read_len=7000, coverage=25.0)
    corr = ctx.purge_palindromes(ctx.scan(ctx.reads_synthetic(spec), K=15, density=0.005, hpc=True), 4, 100)
    for table in (ctx.kminmer_count_first(corr, 4, 0), ctx.kminmer_index(corr, None, 6, ctx.kminmer_count_refined(corr, None, 5, ctx.kminmer_count_first(corr, 4, 0)))):
        rec, vec = table.to_host()
        n = len(rec)
        cuts = [0, 1, n // 3, n // 3, n - 1, n]
        got_r, got_v = [], []
        for a, b in zip(cuts[:-1], cuts[1:]):
            r, v = table.to_host_range(a, b - a)
            got_r.append(r); got_v.append(v)
        assert np.array_equal(np.concatenate(got_r), rec)
        if vec is not None:
            assert np.array_equal(np.concatenate(got_v), vec)
        from metamdbg_amd import capi
        with pytest.raises(capi.MdbgError):
            table.to_host_range(n, 1)


def test_record_files_taken_apart_on_the_device(ctx, orc):
    """mdbg_bytes_* + mdbg_minimizers_from_record_bytes / mdbg_prev_from_record_bytes: read_data_corrected.txt / unitig_data.txt records
    (`u32 n; u8 circular; u32 m[n]`, ReadSelection.hpp:1420-1426) and kminmerData_abundance_prev.txt records handed over as the files'
    BYTES, in pieces, equal the same files parsed on the host and uploaded as arrays -- every misalignment of a record's values (record r
    starts at byte 5 r + 4 off[r]), empty records, the flag bytes; bytes that are not the file the offsets describe are refused."""
    from metamdbg_amd import capi
    rng = np.random.default_rng(606)
    lens = np.concatenate([rng.integers(0, 60, 5000), [0, 0, 1, 70000, 3, 0]])
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    mins = rng.integers(0, 2**32, int(offs[-1]), dtype=np.uint64).astype(np.uint32)
    circ = rng.integers(0, 2, len(lens)).astype(np.uint8)
    raw = bytearray(formats.write_minimizer_reads(mins, offs))
    for r in range(len(lens)):
        raw[5 * r + 4 * int(offs[r]) + 4] = int(circ[r])
    raw = bytes(raw)
    b = ctx.bytes_from_host(raw, piece=100_003)          # pieces that end anywhere
    m, got_circ = ctx.minimizers_from_record_bytes(b, offs, want_circular=True)
    h = m.to_host(full=False)
    assert np.array_equal(h["minimizers"], mins) and np.array_equal(h["offsets"], offs) and np.array_equal(got_circ, circ)
    # the table over it is the table over the uploaded arrays
    rec1, _ = ctx.kminmer_count_first(m, 4, 0).to_host()
    rec2, _ = ctx.kminmer_count_first(ctx.minimizers_from_host(mins, offs), 4, 0).to_host()
    assert np.array_equal(formats.sorted_abundance_records(rec1), formats.sorted_abundance_records(rec2))
    # offsets that do not describe these bytes: same total, another split -> a record's own count disagrees
    bad = offs.copy(); bad[10] += 1
    with pytest.raises(capi.MdbgError):
        ctx.minimizers_from_record_bytes(b, bad)
    with pytest.raises(capi.MdbgError):                  # another size altogether
        ctx.minimizers_from_record_bytes(b, offs[:-1])
    b.free()
    # empty file
    e = ctx.bytes_from_host(b"")
    assert ctx.minimizers_from_record_bytes(e, np.zeros(1, np.uint64)).info() == dict(n_reads=0, n_minimizers=0)
    # a previous table from its file's bytes answers like one from the records
    t = orc.kminmer_count_first(mins % 50, offs, 3, 0)
    recs = orc.table_abundance_records(t).tobytes()
    p1 = ctx.prev_from_records(recs)
    tb = ctx.bytes_from_host(recs, piece=77_777)
    p2 = ctx.prev_from_record_bytes(tb, len(recs) // 20)
    r = formats.parse_abundance_table(recs)
    lo, hi = r["lo"].astype(np.uint64), r["hi"].astype(np.uint64)
    assert np.array_equal(p1.lookup(lo, hi), p2.lookup(lo, hi))
    assert np.array_equal(p2.lookup(lo, hi), np.where(r["abundance"] == 1, p2.lookup(lo, hi), r["abundance"]))
    with pytest.raises(capi.MdbgError):
        ctx.prev_from_record_bytes(tb, len(recs) // 20 + 1)


def test_bad_arguments_are_refused_with_a_code_and_a_message(ctx):
    """include/mdbg_hip.h: every entry returns MDBG_E* with a message in mdbg_last_error instead of crashing, and the context goes on
    working -- null handles, null outputs, parameters outside what the path defines (l > 16, k < 2, a row range past the table, vectors of
    a table that has none, offsets that go backwards, too little room for the census pick)."""
    import ctypes as C
    from metamdbg_amd import capi
    L = capi.lib()
    spec = synth.hifi_spec(300, seed=5, read_len=4000, coverage=20.0)
    reads = ctx.reads_synthetic(spec)
    mins = ctx.scan(reads, K=15, density=0.005, hpc=True)
    corr = ctx.purge_palindromes(mins, 4, 100)
    table = ctx.kminmer_count_first(corr, 4, 0)
    out = C.c_void_p()
    none = C.c_void_p()
    p_ok = capi.ScanParams(15, 0.005, 1, 0.0, None, 0, 1, 0, 0, 0)

    def refused(rc, codes=(-1,)):
        assert rc in codes, rc
        msg = (L.mdbg_last_error(ctx.h) or b"").decode()
        assert msg, "no message"
        return msg

    refused(L.mdbg_scan(ctx.h, none, C.byref(p_ok), C.byref(out)))
    refused(L.mdbg_scan(ctx.h, reads.h, None, C.byref(out)))
    refused(L.mdbg_scan(ctx.h, reads.h, C.byref(p_ok), None))
    for bad_l in (0, 1, 17, 32):
        assert "minimizer_size" in refused(L.mdbg_scan(ctx.h, reads.h, C.byref(capi.ScanParams(bad_l, 0.005, 1, 0.0, None, 0, 1, 0, 0, 0)), C.byref(out)))
    assert "quality_window" in refused(L.mdbg_scan(ctx.h, reads.h, C.byref(capi.ScanParams(15, 0.005, 1, 0.0, None, 0, 1, 7, 0, 0)), C.byref(out)))
    refused(L.mdbg_reads_from_ascii(ctx.h, None, None, None, 5, C.byref(out)))
    refused(L.mdbg_purge_palindromes(ctx.h, mins.h, 1, 100, C.byref(out)))
    refused(L.mdbg_purge_palindromes(ctx.h, none, 4, 100, C.byref(out)))
    refused(L.mdbg_apply_density_threshold(ctx.h, mins.h, C.c_float(0.0), C.byref(out)))
    refused(L.mdbg_kminmer_count_first(ctx.h, corr.h, 1, 0, C.byref(out)))
    refused(L.mdbg_kminmer_count_first(ctx.h, none, 4, 0, C.byref(out)))
    refused(L.mdbg_kminmer_count_refined(ctx.h, corr.h, None, 5, none, C.byref(out)))
    refused(L.mdbg_kminmer_index(ctx.h, corr.h, None, 6, none, C.byref(out)))
    n_rec = table.info()["n_records"]
    buf = np.zeros(20 * 4, np.uint8)
    assert "rows" in refused(L.mdbg_table_to_host_range(ctx.h, table.h, n_rec - 1, 3, buf.ctypes.data_as(C.c_void_p), None))
    refused(L.mdbg_table_checksum(ctx.h, table.h, None))
    off_bad = np.array([0, 5, 3], np.uint64)
    vals = np.arange(8, dtype=np.uint32)
    assert "non-decreasing" in refused(L.mdbg_minimizers_from_host(ctx.h, vals.ctypes.data_as(C.c_void_p), off_bad.ctypes.data_as(C.c_void_p), 2, C.byref(out)))
    # a census pick that does not fit the caller's room: MDBG_ERANGE with the size needed
    room = C.c_uint32(0)
    one = np.zeros(1, np.uint32)
    assert "room" in refused(L.mdbg_repetitive_minimizers(ctx.h, mins.h, one.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(room)), codes=(-5,))
    # a k > firstK+1 table has no vectors to hand out
    t5 = ctx.kminmer_count_refined(corr, None, 5, table)
    t6 = ctx.kminmer_index(corr, None, 6, t5)
    n6 = t6.info()["n_records"]
    vec = np.zeros(max(1, n6) * 6, np.uint32)
    rec = np.zeros(max(1, n6) * 20, np.uint8)
    assert "no vectors" in refused(L.mdbg_table_to_host(ctx.h, t6.h, rec.ctypes.data_as(C.c_void_p), vec.ctypes.data_as(C.c_void_p)))
    assert L.mdbg_set_option(ctx.h, b"no_such_option", 1) != 0
    # ... and the context is as good as before
    again = ctx.kminmer_count_first(ctx.purge_palindromes(ctx.scan(reads, K=15, density=0.005, hpc=True), 4, 100), 4, 0)
    assert again.checksum() == table.checksum() and again.info()["n_records"] == n_rec
