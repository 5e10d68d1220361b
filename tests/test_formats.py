"""metamdbg_amd/formats.py (harness side: file layouts and the order-independent digests the 1 M-read fixtures are compared through)."""
from __future__ import annotations

import struct

import numpy as np

from metamdbg_amd import formats


def _scan_output(rng, n_reads):
    cnt = rng.integers(0, 60, n_reads)
    cnt[n_reads // 3] = 0
    cnt[-1] = 0
    offs = np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint64)
    t = int(offs[-1])
    h = dict(offsets=offs, minimizers=rng.integers(0, 2**32, t, dtype=np.uint64).astype(np.uint32), pos=rng.integers(0, 2**20, t).astype(np.uint32),
             dir=rng.integers(0, 2, t).astype(np.uint8), qual=rng.integers(0, 60, t).astype(np.uint8), mean_quality=np.full(n_reads, np.nan, np.float32),
             read_length=rng.integers(0, 2**20, n_reads).astype(np.uint32))
    h["mean_quality"][3] = 12.5
    return h


def test_read_data_init_layout_matches_the_record_by_record_form():
    """u32 n; u8 circ = 0; u32 m[n]; u32 pos[n]; u8 dir[n]; u8 qual[n]; f32 meanQ; u32 len (readSelection/ReadSelection.hpp:415-467): the
    vectorised builder against the obvious loop, empty reads and NaN mean qualities included; and the parser reads it back."""
    h = _scan_output(np.random.default_rng(1), 2000)
    offs, parts = h["offsets"], []
    mq = np.asarray(h["mean_quality"], dtype="<f4")
    for r in range(len(offs) - 1):
        a, b = int(offs[r]), int(offs[r + 1])
        parts += [struct.pack("<IB", b - a, 0), h["minimizers"][a:b].astype("<u4").tobytes(), h["pos"][a:b].astype("<u4").tobytes(),
                  h["dir"][a:b].tobytes(), h["qual"][a:b].tobytes(), mq[r:r + 1].tobytes(), struct.pack("<I", int(h["read_length"][r]))]
    raw = formats.build_read_data_init(h)
    assert raw == b"".join(parts)
    back = formats.parse_read_data_init(raw)
    assert len(back) == 2000 and all(np.array_equal(back[r]["minimizers"], h["minimizers"][int(offs[r]): int(offs[r + 1])]) for r in (0, 7, 1999))
    assert formats.build_read_data_init(dict(h, offsets=offs[:1], read_length=h["read_length"][:0], mean_quality=h["mean_quality"][:0])) == b""


def test_multiset_comparison_and_digest_of_minimizer_reads():
    """read_data_corrected.txt is written in thread order by the reference: equal as multisets of reads whatever the order, different as soon as
    one value, one length or the order inside one read differs."""
    rng = np.random.default_rng(2)
    h = _scan_output(rng, 3000)
    m, o = h["minimizers"], h["offsets"]
    perm = rng.permutation(3000)
    parts = [m[int(o[i]): int(o[i + 1])] for i in perm]
    m2 = np.concatenate(parts)
    o2 = np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.uint64)
    assert formats.minimizer_reads_equal_as_multisets(m, o, m2, o2)
    assert formats.minimizer_reads_digest(m, o) == formats.minimizer_reads_digest(m2, o2)
    m3 = m2.copy(); m3[100] ^= 4
    assert not formats.minimizer_reads_equal_as_multisets(m, o, m3, o2) and formats.minimizer_reads_digest(m3, o2) != formats.minimizer_reads_digest(m, o)
    r = int(np.flatnonzero(np.diff(o.astype(np.int64)) >= 2)[0]); a = int(o[r])
    if m[a] != m[a + 1]:
        m4 = m.copy(); m4[a], m4[a + 1] = m[a + 1], m[a]
        assert not formats.minimizer_reads_equal_as_multisets(m, o, m4, o) and formats.minimizer_reads_digest(m4, o) != formats.minimizer_reads_digest(m, o)
    o5 = o.copy(); o5[10] += 1          # one minimizer moves from read 10 to read 9
    assert not formats.minimizer_reads_equal_as_multisets(m, o, m, o5)
    # the file form round-trips
    assert formats.parse_minimizer_reads(formats.write_minimizer_reads(m, o))[0].tolist() == m.tolist()


def test_table_digests_ignore_record_order():
    rng = np.random.default_rng(3)
    rec = np.zeros(500, dtype=formats.ABUNDANCE_DTYPE)
    rec["lo"], rec["hi"], rec["abundance"] = rng.integers(0, 2**63, 500), rng.integers(0, 2**63, 500), rng.integers(1, 90, 500)
    vec = rng.integers(0, 2**32, (500, 4), dtype=np.uint64).astype("<u4")
    p = rng.permutation(500)
    assert formats.table_digests(rec, vec, 4) == formats.table_digests(rec[p], vec[p], 4)
    rec2 = rec.copy(); rec2["abundance"][7] += 1
    assert formats.table_digests(rec2, vec, 4)["abundance_sorted_sha256"] != formats.table_digests(rec, vec, 4)["abundance_sorted_sha256"]
