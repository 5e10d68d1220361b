"""CPU test of the C++ host tool's parallel FASTA/FASTQ reader against its sequential reader."""
from __future__ import annotations

import gzip
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("hostfeed") / "test_hostfeed")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", os.path.join(ROOT, "tests", "host", "test_hostfeed.cpp"), "-o", out,
                    "-lz", "-lpthread"], check=True)
    return out


def _bgzf(data: bytes, block: int = 0xff00, level: int = 6, eof_marker: bool = True) -> bytes:
    """BGZF as htslib writes it (SAM spec 4.1): gzip members of <= 64 KB with the 'BC' extra subfield = block size - 1."""
    import struct
    import zlib
    out = bytearray()
    chunks = [data[o:o + block] for o in range(0, len(data), block)] + ([b""] if eof_marker else [])
    for c in chunks:
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        payload = co.compress(c) + co.flush()
        bsize = 12 + 6 + len(payload) + 8
        out += struct.pack("<BBBBIBBH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6) + b"BC" + struct.pack("<HH", 2, bsize - 1)
        out += payload + struct.pack("<II", zlib.crc32(c) & 0xffffffff, len(c))
    return bytes(out)


def _rand_seq(rng, n):
    return bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)])


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    d = tmp_path_factory.mktemp("reads")
    rng = np.random.default_rng(3)
    out = {}
    # single-line FASTA
    p = str(d / "single.fasta")
    with open(p, "wb") as f:
        for i in range(400):
            f.write(b">r%d desc\n" % i + _rand_seq(rng, int(rng.integers(1, 5000))) + b"\n")
    out["single"] = p
    # wrapped FASTA with CRLF and an empty read, no trailing newline
    p = str(d / "wrapped.fasta")
    with open(p, "wb") as f:
        for i in range(300):
            s = _rand_seq(rng, int(rng.integers(0, 3000)))
            f.write(b">w%d\r\n" % i)
            for o in range(0, len(s), 60):
                f.write(s[o:o + 60] + b"\r\n")
        f.write(b">last\nACGTACGT")
    out["wrapped"] = p
    # FASTQ with '@' and '+' inside quality strings
    p = str(d / "reads.fastq")
    with open(p, "wb") as f:
        for i in range(350):
            n = int(rng.integers(1, 4000))
            q = (rng.integers(0, 61, n) + 33).astype(np.uint8)
            q[0] = ord("@") if i % 3 == 0 else q[0]
            q[-1] = ord("+") if i % 5 == 0 else q[-1]
            f.write(b"@q%d\n" % i + _rand_seq(rng, n) + b"\n+\n" + bytes(q) + b"\n")
    out["fastq"] = p
    # FASTA with N / lower-case n in a few reads: those chunks must come through as ASCII, the others packed
    p = str(d / "with_n.fasta")
    with open(p, "wb") as f:
        for i in range(600):
            sq = bytearray(_rand_seq(rng, int(rng.integers(200, 3000))))
            if i in (17, 400, 401):
                sq[len(sq) // 2] = ord("N") if i != 401 else ord("n")
            f.write(b">n%d\n" % i + bytes(sq) + b"\n")
    out["with_n"] = p
    # a soft-masked / lower-case FASTA, wrapped lines: every read is odd -- such chunks are delivered as ASCII
    p = str(d / "lower.fasta")
    with open(p, "wb") as f:
        for i in range(500):
            sq = _rand_seq(rng, int(rng.integers(200, 3000))).lower()
            f.write(b">l%d\n" % i + b"\n".join(sq[j:j + 70] for j in range(0, len(sq), 70)) + b"\n")
    out["lower"] = p
    # gzipped FASTA
    p = str(d / "z.fasta.gz")
    with gzip.open(p, "wb") as f:
        for i in range(200):
            f.write(b">z%d\n" % i + _rand_seq(rng, int(rng.integers(1, 4000))) + b"\n")
    out["gz"] = p
    # gzipped 4-line FASTQ (inflated slabs go to the parallel workers), '@' / '+' at the ends of quality strings
    p = str(d / "z.fastq.gz")
    with gzip.open(p, "wb") as f:
        for i in range(300):
            n = int(rng.integers(1, 4000))
            q = (rng.integers(0, 61, n) + 33).astype(np.uint8)
            q[0] = ord("@") if i % 3 == 0 else q[0]
            q[-1] = ord("+") if i % 4 == 0 else q[-1]
            f.write(b"@g%d\n" % i + _rand_seq(rng, n) + b"\n+g%d\n" % i + bytes(q) + b"\n")
    out["gz_fastq"] = p
    # gzipped wrapped FASTA, two gzip members, no trailing newline
    p = str(d / "zw.fasta.gz")
    with open(p, "wb") as raw:
        for member in range(2):
            with gzip.GzipFile(fileobj=raw, mode="wb") as f:
                for i in range(150):
                    s = _rand_seq(rng, int(rng.integers(0, 3000)))
                    f.write(b">m%d_%d\n" % (member, i))
                    for o in range(0, len(s), 80):
                        f.write(s[o:o + 80] + b"\n")
                if member == 1:
                    f.write(b">last\nACGTTGCA")
    out["gz_wrapped"] = p
    # gzipped multi-line FASTQ: the sequential reader's job
    p = str(d / "zm.fastq.gz")
    with gzip.open(p, "wb") as f:
        for i in range(120):
            n = int(rng.integers(100, 1500))
            s_, q_ = _rand_seq(rng, n), bytes((rng.integers(0, 40, n) + 33).astype(np.uint8))
            f.write(b"@ml%d\n" % i)
            for o in range(0, n, 70):
                f.write(s_[o:o + 70] + b"\n")
            f.write(b"+\n")
            for o in range(0, n, 70):
                f.write(q_[o:o + 70] + b"\n")
    out["gz_multiline_fastq"] = p
    # the same records uncompressed: kseq accepts multi-line FASTQ (Commons.hpp:82), so must the memory-mapped path
    p = str(d / "m.fastq")
    with gzip.open(out["gz_multiline_fastq"], "rb") as f, open(p, "wb") as g:
        g.write(f.read())
    out["multiline_fastq"] = p
    # BGZF (samtools fastq / bam2fastq / bgzip output): FASTQ, small blocks so that records straddle many of them
    recs = bytearray()
    for i in range(500):
        n = int(rng.integers(1, 4000))
        q = (rng.integers(0, 61, n) + 33).astype(np.uint8)
        q[0] = ord("@") if i % 3 == 0 else q[0]
        recs += b"@b%d\n" % i + _rand_seq(rng, n) + b"\n+\n" + bytes(q) + b"\n"
    p = str(d / "b.fastq.gz")
    open(p, "wb").write(_bgzf(bytes(recs), block=3000))
    out["bgzf_fastq"] = p
    p = str(d / "b64k.fastq.gz")
    open(p, "wb").write(_bgzf(bytes(recs), eof_marker=False))
    out["bgzf_fastq_64k"] = p
    # BGZF FASTA followed by an ordinary gzip member: not pure BGZF, must take the gzread path
    p = str(d / "mixed.fasta.gz")
    fa = b"".join(b">x%d\n" % i + _rand_seq(rng, int(rng.integers(1, 3000))) + b"\n" for i in range(100))
    open(p, "wb").write(_bgzf(fa[: len(fa) // 2], eof_marker=False) + gzip.compress(fa[len(fa) // 2:]))
    out["bgzf_then_gzip"] = p
    return out


@pytest.mark.parametrize("chunk", [10000, 50000, 1 << 22])
@pytest.mark.parametrize("threads", [1, 4])
def test_parallel_reader_matches_sequential(exe, files, chunk, threads):
    for key in ("single", "wrapped", "fastq", "gz", "gz_fastq", "gz_wrapped", "gz_multiline_fastq", "multiline_fastq", "bgzf_fastq",
                "bgzf_fastq_64k", "bgzf_then_gzip"):
        r = subprocess.run([exe, str(chunk), str(threads), "0", files[key]], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, (key, r.stderr)
    r = subprocess.run([exe, str(chunk), str(threads), "0", files["single"], files["gz"], files["fastq"], files["gz_fastq"],
                        files["wrapped"], files["gz_multiline_fastq"], files["bgzf_fastq"], files["multiline_fastq"], files["gz_wrapped"]],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr


def test_per_file_read_cap(exe, files):
    r = subprocess.run([exe, "20000", "3", "100", files["single"], files["fastq"], files["gz"]], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout.startswith("ok 303 reads")      # 101 reads of each file (readIndexPerDataset > maxReads stops a file)


def test_packed_and_ascii_batches(exe, files):
    """Workers pack chunks to 2 bits; a read with anything but upper-case ACGT is packed all the same (code (c >> 1) & 3) and carried
    a second time as characters (ReadBatch::odd -> mdbg_reads_mark_ascii); only a chunk in which such reads are many is delivered as
    ASCII (the harness checks that exactly the reads that need it are listed, with their characters)."""
    r = subprocess.run([exe, "20000", "3", "0", files["with_n"]], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    toks = r.stdout.split()          # ok <reads> reads <batches> batches <packed> packed <odd> odd
    n_batches, n_packed, n_odd = int(toks[3]), int(toks[5]), int(toks[7])
    assert toks[1] == "600" and n_packed == n_batches and 0 < n_odd <= 3
    # many odd reads (a lower-case file): the chunks come as characters once enough of their reads turned out odd
    r = subprocess.run([exe, "400000", "3", "0", files["lower"]], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    toks = r.stdout.split()
    assert toks[1] == "500" and int(toks[5]) < int(toks[3])
    r = subprocess.run([exe, "20000", "3", "0", files["single"]], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.split()[3] == r.stdout.split()[5]          # all packed
    r = subprocess.run([exe, "20000", "3", "0", files["single"]], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, MDBG_HOST_NO_PACK="1"))
    assert r.returncode == 0 and r.stdout.split()[5] == "0"
    # the portable (no BMI2 pext) packing path
    for key in ("single", "wrapped", "fastq", "with_n"):
        r = subprocess.run([exe, "10000", "2", "0", files[key]], capture_output=True, text=True, timeout=120,
                           env=dict(os.environ, MDBG_HOST_NO_BMI2="1"))
        assert r.returncode == 0 and int(r.stdout.split()[5]) > 0, (key, r.stderr)


def test_bgzf_corruption_is_reported(exe, files, tmp_path):
    """A flipped byte inside a BGZF block fails the block's CRC (or the inflate) and surfaces as an error, not as wrong reads."""
    raw = bytearray(open(files["bgzf_fastq_64k"], "rb").read())
    raw[len(raw) // 2] ^= 0x55
    bad = str(tmp_path / "bad.fastq.gz")
    open(bad, "wb").write(bytes(raw))
    # (the pool inflates the blocks straight into the slab; MDBG_HOST_BGZF_COPY=1: into buffers of its own, copied from there)
    for mode in ({}, {"MDBG_HOST_BGZF_COPY": "1"}, {"MDBG_HOST_ZLIB_INFLATE": "1"}):
        r = subprocess.run([exe, str(1 << 20), "3", "0", bad], capture_output=True, text=True, timeout=120,
                           env=dict(os.environ, MDBG_TEST_DRAIN_ONLY="1", **mode))
        assert r.returncode == 3, (mode, r.returncode, r.stdout, r.stderr)


@pytest.mark.parametrize("chunk", [10000, 200000, 1 << 22])
def test_bgzf_copy_mode_matches(exe, files, chunk):
    for key in ("bgzf_fastq", "bgzf_fastq_64k", "bgzf_then_gzip"):
        r = subprocess.run([exe, str(chunk), "4", "0", files[key]], capture_output=True, text=True, timeout=120,
                           env=dict(os.environ, MDBG_HOST_BGZF_COPY="1"))
        assert r.returncode == 0, (key, r.stderr)


def test_gzip_damage_and_zlib_switch(exe, files, tmp_path):
    """Ordinary gzip: a flipped byte or a truncated file is an error (CRC-32 / length of every member are checked behind
    the decoder), trailing garbage behind the last member is ignored as gzread does, and MDBG_HOST_ZLIB_INFLATE=1 decodes
    the same reads with zlib."""
    raw = open(files["gz_fastq"], "rb").read()
    bad = bytearray(raw)
    bad[len(bad) // 2] ^= 0x10
    p = str(tmp_path / "flipped.fastq.gz")
    open(p, "wb").write(bytes(bad))
    drain = dict(os.environ, MDBG_TEST_DRAIN_ONLY="1")     # exit code 3 = the feeder ended with an error, 0 = it delivered "all" reads
    modes = ({}, {"MDBG_HOST_ZLIB_INFLATE": "1"}, {"MDBG_HOST_GZIP_THREADS": "4", "MDBG_HOST_GZIP_CHUNK": "65536"})
    for mode in modes:
        r = subprocess.run([exe, str(1 << 20), "3", "0", p], capture_output=True, text=True, timeout=120, env=dict(drain, **mode))
        assert r.returncode == 3, (mode, r.returncode, r.stdout, r.stderr)
    p = str(tmp_path / "truncated.fastq.gz")
    open(p, "wb").write(raw[: len(raw) * 2 // 3])
    for mode in modes:
        r = subprocess.run([exe, str(1 << 20), "3", "0", p], capture_output=True, text=True, timeout=120, env=dict(drain, **mode))
        assert r.returncode == 3, (mode, r.returncode, r.stdout, r.stderr)
    # damage inside the first slab sends the file to the sequential reader (no 4-line records to be seen): still an error,
    # not a silently shortened read set
    bad = bytearray(raw)
    bad[len(bad) // 50] ^= 0x04
    p = str(tmp_path / "early.fastq.gz")
    open(p, "wb").write(bytes(bad))
    for mode in modes:
        r = subprocess.run([exe, str(1 << 20), "3", "0", p], capture_output=True, text=True, timeout=120, env=dict(drain, **mode))
        assert r.returncode == 3, (mode, r.returncode, r.stdout, r.stderr)
    p = str(tmp_path / "garbage_tail.fastq.gz")
    open(p, "wb").write(raw + b"\x00" * 37 + b"not gzip")
    r = subprocess.run([exe, str(1 << 20), "3", "0", p], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    for key in ("gz", "gz_fastq", "gz_wrapped", "bgzf_fastq"):
        r = subprocess.run([exe, "50000", "3", "0", files[key]], capture_output=True, text=True, timeout=120,
                           env=dict(os.environ, MDBG_HOST_ZLIB_INFLATE="1"))
        assert r.returncode == 0, (key, r.stderr)


def test_inflate_against_zlib(tmp_path):
    """metamdbg_amd/host/inflate.hpp vs zlib: every block type, level and strategy, output room cut at random places, and
    damaged streams -- built with the address and undefined-behaviour sanitizers."""
    out = str(tmp_path / "test_inflate")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                    os.path.join(ROOT, "tests", "host", "test_inflate.cpp"), "-o", out, "-lz"], check=True)
    r = subprocess.run([out, "11", "30"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr + r.stdout
    assert r.stdout.startswith("ok 30 streams")


def test_record_files_indexed_on_several_threads(tmp_path):
    """metamdbg_amd/host/records.hpp -- read_data_corrected.txt / unitig_data.txt walked by several threads that GUESS a record start
    in their chunk and are joined exactly -- against the serial walk: random files, values that read as record headers (false guesses),
    records longer than a chunk, empty records, all-zero files, truncation; 1 - 64 threads; address and UB sanitizers."""
    out = str(tmp_path / "test_records")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                    os.path.join(ROOT, "tests", "host", "test_records.cpp"), "-o", out, "-lpthread"], check=True)
    r = subprocess.run([out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr + r.stdout
    assert r.stdout.startswith("ok: 180 cases")


@pytest.fixture(scope="module")
def big_gz(tmp_path_factory):
    """A few MB of gzip so that the several-threads decoder has chunks to cut: FASTQ at three levels, three members with
    trailing zeros, FASTA."""
    d = tmp_path_factory.mktemp("biggz")
    rng = np.random.default_rng(9)
    recs = bytearray()
    for i in range(2500):
        n = int(rng.integers(1, 3000))
        q = (rng.integers(0, 40, n) + 33).astype(np.uint8)
        recs += b"@p%d\n" % i + _rand_seq(rng, n) + b"\n+\n" + bytes(q) + b"\n"
    recs = bytes(recs)
    out = {}
    for lvl in (1, 6, 9):
        out[f"l{lvl}"] = str(d / f"l{lvl}.fastq.gz")
        open(out[f"l{lvl}"], "wb").write(gzip.compress(recs, lvl))
    c1 = recs.rfind(b"\n@p", 0, len(recs) // 3) + 1
    c2 = recs.rfind(b"\n@p", 0, 2 * len(recs) // 3) + 1
    out["members"] = str(d / "m3.fastq.gz")
    open(out["members"], "wb").write(gzip.compress(recs[:c1], 6) + gzip.compress(recs[c1:c2], 1) + gzip.compress(recs[c2:], 9) + b"\0" * 64)
    fa = b"".join(b">f%d\n" % i + _rand_seq(rng, int(rng.integers(1, 20000))) + b"\n" for i in range(800))
    out["fasta"] = str(d / "f.fasta.gz")
    open(out["fasta"], "wb").write(gzip.compress(fa, 6))
    # stored blocks only (gzip -0): nothing to cut at, one thread must take it; and a stored member between compressed ones
    out["stored"] = str(d / "s0.fastq.gz")
    open(out["stored"], "wb").write(gzip.compress(recs, 0))
    out["stored_between"] = str(d / "s1.fastq.gz")
    open(out["stored_between"], "wb").write(gzip.compress(recs[:c1], 6) + gzip.compress(recs[c1:c2], 0) + gzip.compress(recs[c2:], 6))
    return out


@pytest.mark.parametrize("gzchunk", [65536, 150000, 700000])
def test_one_gzip_stream_on_several_threads(exe, big_gz, gzchunk):
    """gzip_parallel.hpp: chunks decoded without their window and resolved afterwards must give exactly the reads of the
    sequential reader -- blocks longer than a chunk, members ending inside chunks, trailing bytes."""
    env = dict(os.environ, MDBG_HOST_GZIP_THREADS="4", MDBG_HOST_GZIP_CHUNK=str(gzchunk))
    for key, path in big_gz.items():
        r = subprocess.run([exe, "300000", "3", "0", path], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, (key, r.stderr)
    r = subprocess.run([exe, "300000", "2", "0", big_gz["members"], big_gz["fasta"], big_gz["l1"]], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr


def test_damaged_gzip_on_several_threads(exe, big_gz, tmp_path):
    raw = open(big_gz["l6"], "rb").read()
    rng = np.random.default_rng(4)
    env = dict(os.environ, MDBG_HOST_GZIP_THREADS="4", MDBG_HOST_GZIP_CHUNK="150000", MDBG_TEST_DRAIN_ONLY="1")
    for t in range(12):
        bad = bytearray(raw)
        if t % 4 == 3:
            bad = bad[: int(rng.integers(len(bad) // 4, len(bad) - 1))]
        else:
            for _ in range(int(rng.integers(1, 4))):
                bad[int(rng.integers(100, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        p = str(tmp_path / "bad.fastq.gz")
        open(p, "wb").write(bytes(bad))
        r = subprocess.run([exe, "300000", "3", "0", p], capture_output=True, text=True, timeout=120, env=env)
        assert r.returncode == 3, (t, r.returncode, r.stdout, r.stderr)


@pytest.mark.parametrize("gzchunk", [65536, 200000, 1000000])
def test_many_gzip_members_on_several_threads(exe, tmp_path, gzchunk):
    """The chunks are cut over the FILE, whatever members it is made of: thousands of tiny members, members of a few MB, stored
    (gzip -0) members in between, garbage behind the last one -- the same reads as the sequential reader, at every chunk size."""
    rng = np.random.default_rng(18)
    fa = b"".join(b">m%d\n" % i + _rand_seq(rng, int(rng.integers(1, 30000))) + b"\n" for i in range(1200))

    def members(sizes, levels, tail=b""):
        parts, o = [], 0
        while o < len(fa):
            e = fa.find(b"\n>", o + int(rng.choice(sizes)))
            e = len(fa) if e < 0 else e + 1
            parts.append(gzip.compress(fa[o:e], int(rng.choice(levels))))
            o = e
        return b"".join(parts) + tail
    files = []
    for name, data in (("tiny", members([100, 3000, 60000], (1, 6))), ("mid", members([200000, 3000000], (1, 6), b"\0\0garbage")),
                       ("stored", members([100, 5000, 2000000], (0, 1, 6)))):
        p = str(tmp_path / f"{name}.fasta.gz")
        open(p, "wb").write(data)
        files.append(p)
    env = dict(os.environ, MDBG_HOST_GZIP_THREADS="6", MDBG_HOST_GZIP_CHUNK=str(gzchunk))
    for chunk in ("200000", str(1 << 22)):
        r = subprocess.run([exe, chunk, "4", "0", *files], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr
        assert r.stdout.startswith("ok 3600 reads")


def test_damaged_and_truncated_inputs_under_load(exe, big_gz, tmp_path):
    """Damaged and truncated gzip / BGZF files with sixteen readers at once on the machine: an error every time, never a hang.  (A run
    that threw at a damaged chunk once forgot to tell the pool about the pieces it had already handed out and then waited for them:
    one run in eight, and only on a busy machine -- which is where this test puts it.)"""
    import concurrent.futures as cf
    rng = np.random.default_rng(4)
    raws = {"gz": open(big_gz["l6"], "rb").read(), "bgzf": _bgzf(gzip.decompress(open(big_gz["l6"], "rb").read()), block=20000)}
    cases = []
    for kind, raw in raws.items():
        for t in range(8):
            bad = bytearray(raw)
            if t % 2:
                bad = bad[: int(rng.integers(len(bad) // 4, len(bad) - 1))]
            else:
                for _ in range(int(rng.integers(1, 4))):
                    bad[int(rng.integers(100, len(bad)))] ^= 1 << int(rng.integers(0, 8))
            p = str(tmp_path / f"bad_{kind}_{t}.fastq.gz")
            open(p, "wb").write(bytes(bad))
            cases.append(p)
    env = dict(os.environ, MDBG_HOST_GZIP_THREADS="4", MDBG_HOST_GZIP_CHUNK="150000", MDBG_TEST_DRAIN_ONLY="1")

    def one(p):
        try:
            return p, subprocess.run([exe, "300000", "3", "0", p], capture_output=True, text=True, timeout=60, env=env).returncode
        except subprocess.TimeoutExpired:
            return p, "hang"
    with cf.ThreadPoolExecutor(16) as ex:
        res = list(ex.map(one, cases * 6))
    assert all(rc == 3 for _, rc in res), [(os.path.basename(p), rc) for p, rc in res if rc != 3]


def test_reads_nearly_as_long_as_a_batch(exe, tmp_path):
    """Reads of 0.5 to 1.5 Mbp and of 5 to 5.9 Mbp in turn, batches of 6 MB: a cut behind a short read in front of a long one leaves over
    more (the long read's start and 4 MB of look-ahead: 9 MB) than the 8 MB kept in front of the next slab's text, and the slab is rebuilt
    at its own size -- BGZF and gzip on several threads, the paths that fill slabs in place."""
    rng = np.random.default_rng(31)
    pool = _rand_seq(rng, 6_000_000)
    fa = b"".join(b">long%d\n" % i + pool[int(rng.integers(0, 100_000)):][: int(rng.integers(5_000_000, 5_900_000) if i % 2 else rng.integers(500_000, 1_500_000))] + b"\n"
                  for i in range(10))
    files = []
    for name, data in (("long.bgzf.fasta.gz", _bgzf(fa, level=1)), ("long.fasta.gz", gzip.compress(fa, 1))):
        p = str(tmp_path / name)
        open(p, "wb").write(data)
        files.append(p)
    env = dict(os.environ, MDBG_HOST_GZIP_THREADS="6", MDBG_HOST_GZIP_CHUNK="1000000")
    for f in files:
        r = subprocess.run([exe, "6000000", "4", "0", f], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0 and r.stdout.startswith("ok 10 reads"), (f, r.stdout, r.stderr)
