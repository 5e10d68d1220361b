#!/usr/bin/env python3
"""Steady-state end-to-end measurement of the C++ drop-in (mdbg_tool readSelection + graph --firstpass) from files in /dev/shm
(page cache), tool only, with the tool's own phase trace (MDBG_TRACE=1):

    python tools/e2e_steady.py --reads 5000000 --fastq-reads 2000000 --gz-reads 1000000 --threads 32,64 --out gpurun_out/e2e.json

  fasta   n x 10 kb synthetic HiFi reads as plain FASTA (50 Gbp at the default)
  fastq   the same preset with qualities (1 byte per base more over the link; mean read quality, per-minimizer minimum)
  gzip    the first --gz-reads reads of the FASTA as an ordinary multi-member gzip stream (gzip -1, members compressed by a process
          pool and concatenated, as `cat a.gz b.gz` would): the several-threads decoder of host/gzip_parallel.hpp at size
The reference is not run here (bench.py times it on 10 Gbp: cpu_baseline); this tool never touches oracle/."""
from __future__ import annotations

import argparse
import dataclasses
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TOOL = os.path.join(ROOT, "metamdbg_amd", "bin", "mdbg_tool")


def write_reads(path, ctx, spec, n, with_quality):
    reads = ctx.reads_synthetic(spec)
    t0 = time.perf_counter()
    with open(path, "wb") as f:
        step = 50_000
        for r0 in range(0, n, step):
            c = min(step, n - r0)
            b, o = reads.export_ascii(r0, c)
            if with_quality:
                q = reads.export_qualities(r0, c)
                f.write(b"".join(b"@r%d\n%s\n+\n%s\n" % (r0 + r, b[int(o[r]):int(o[r + 1])].tobytes(), q[int(o[r]):int(o[r + 1])].tobytes()) for r in range(c)))
            else:
                f.write(b"".join(b">r%d\n%s\n" % (r0 + r, b[int(o[r]):int(o[r + 1])].tobytes()) for r in range(c)))
    reads.free()
    return time.perf_counter() - t0


def _bgzf_part(args):
    """A piece of the text as BGZF blocks (htslib's blocked gzip, SAM spec 4.1): gzip members of <= 64 KB carrying their size in a 'BC' extra field."""
    import struct
    path, a, b = args
    with open(path, "rb") as f:
        f.seek(a)
        raw = f.read(b - a)
    out = bytearray()
    for o in range(0, len(raw), 0xff00):
        c = raw[o:o + 0xff00]
        co = zlib.compressobj(1, zlib.DEFLATED, -15)
        payload = co.compress(c) + co.flush()
        out += struct.pack("<BBBBIBBH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(payload) + 8 - 1)
        out += payload + struct.pack("<II", zlib.crc32(c) & 0xffffffff, len(c))
    return bytes(out)


def _gz_member(args):
    path, a, b = args
    with open(path, "rb") as f:
        f.seek(a)
        raw = f.read(b - a)
    co = zlib.compressobj(1, zlib.DEFLATED, 31)
    return co.compress(raw) + co.flush()


def _deflate_piece(args):
    """A piece of ONE deflate stream, the way pigz makes them: raw deflate, ended on a byte border by a sync flush (the last piece:
    finished), so that the pieces of all workers concatenated are a single stream; with the piece's CRC-32 and length."""
    path, a, b, last = args
    with open(path, "rb") as f:
        f.seek(a)
        raw = f.read(b - a)
    co = zlib.compressobj(1, zlib.DEFLATED, -15)
    return co.compress(raw) + co.flush(zlib.Z_FINISH if last else zlib.Z_SYNC_FLUSH), zlib.crc32(raw), len(raw)


def _crc32_combine(crc1, crc2, len2):
    """zlib's crc32_combine (not exposed by Python): the CRC-32 of A + B from those of A and B and len(B), by GF(2) matrix powers."""
    def times(mat, vec):
        s, i = 0, 0
        while vec:
            if vec & 1:
                s ^= mat[i]
            vec >>= 1
            i += 1
        return s

    def square(mat):
        return [times(mat, mat[n]) for n in range(32)]
    if len2 <= 0:
        return crc1
    odd = [0xedb88320] + [1 << n for n in range(31)]      # the operator for one zero bit
    even = square(odd)                                       # two
    odd = square(even)                                       # four
    while True:
        even = square(odd)
        if len2 & 1:
            crc1 = times(even, crc1)
        len2 >>= 1
        if not len2:
            break
        odd = square(even)
        if len2 & 1:
            crc1 = times(odd, crc1)
        len2 >>= 1
        if not len2:
            break
    return crc1 ^ crc2


def run_tool(tmp_parent, inputs, threads, P, trace=True, rs_args=(), graph_args=("--min-abundance", "0", "--firstpass"), extra_env=None):
    from metamdbg_amd import formats
    tmp = os.path.join(tmp_parent, "tmp")
    shutil.rmtree(tmp_parent, ignore_errors=True)
    for d in ("", "filter", "smallContigs", "checkpoints"):
        os.makedirs(os.path.join(tmp, d), exist_ok=True)
    P.save(os.path.join(tmp, "parameters.gz"))
    open(os.path.join(tmp, "input.txt"), "w").write("\n".join(inputs) + "\n")
    env = dict(os.environ, MDBG_TRACE="1") if trace else dict(os.environ)
    env.update(extra_env or {})
    t0 = time.perf_counter()
    r1 = subprocess.run([TOOL, "readSelection", tmp, tmp + "/read_data_init.txt", tmp + "/input.txt", "--threads", str(threads),
                         "--min-read-quality", "0.000000", *rs_args], capture_output=True, text=True, env=env, timeout=240)
    t1 = time.perf_counter()
    assert r1.returncode == 0, r1.stderr[-1000:]
    r2 = subprocess.run([TOOL, "graph", tmp, "--threads", str(threads), *graph_args], capture_output=True, text=True, env=env, timeout=240)
    t2 = time.perf_counter()
    assert r2.returncode == 0, r2.stderr[-1000:]
    sizes = {n: os.path.getsize(os.path.join(tmp, n)) for n in ("read_data_init.txt", "read_data_corrected.txt", "kminmerData_abundance.txt", "kminmerData_min.txt")}
    one = None
    if graph_args == ("--min-abundance", "0", "--firstpass") and os.environ.get("MDBG_E2E_ASM_STEP", "1") == "1":
        # ... and the two commands as ONE process (mdbg_tool asmStep: one context, the corrected minimizers handed over on the device)
        tmp1 = os.path.join(tmp_parent, "one", "tmp")
        for d in ("", "filter", "smallContigs", "checkpoints"):
            os.makedirs(os.path.join(tmp1, d), exist_ok=True)
        P.save(os.path.join(tmp1, "parameters.gz"))
        open(os.path.join(tmp1, "input.txt"), "w").write("\n".join(inputs) + "\n")
        t3 = time.perf_counter()
        r3 = subprocess.run([TOOL, "asmStep", tmp1, tmp1 + "/read_data_init.txt", tmp1 + "/input.txt", "--threads", str(threads), "--min-read-quality", "0.000000",
                             *rs_args, "--min-abundance", "0"], capture_output=True, text=True, env=env, timeout=240)
        t4 = time.perf_counter()
        assert r3.returncode == 0, r3.stderr[-1000:]
        same = all(os.path.getsize(os.path.join(tmp1, n)) == v for n, v in sizes.items()) and \
            open(os.path.join(tmp1, "read_data_init.txt"), "rb").read(1 << 24) == open(os.path.join(tmp, "read_data_init.txt"), "rb").read(1 << 24)
        one = {"asm_step_s": t4 - t3, "output_sizes_equal_and_first_16_MB_of_read_data_init_equal": bool(same),
               "trace": [ln.strip() for ln in r3.stderr.splitlines() if "[mdbg_tool]" in ln]}
        shutil.rmtree(os.path.join(tmp_parent, "one"), ignore_errors=True)
    return {"read_selection_s": t1 - t0, "graph_s": t2 - t1, "total_s": t2 - t0, "asm_step": one,
            "trace_read_selection": [ln.strip() for ln in r1.stderr.splitlines() if "[mdbg_tool]" in ln],
            "trace_graph": [ln.strip() for ln in r2.stderr.splitlines() if "[mdbg_tool]" in ln], "output_bytes": sizes}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=5_000_000)
    ap.add_argument("--fastq-reads", type=int, default=2_000_000)
    ap.add_argument("--gz-reads", type=int, default=1_000_000)
    ap.add_argument("--threads", default="32,64")
    ap.add_argument("--dir", default="/dev/shm")
    ap.add_argument("--out", default="")
    ap.add_argument("--ont-reads", type=int, default=0, help="ONT preset: n x 20 kb reads with qualities as FASTQ, no HPC, census + --skip-correction")
    ap.add_argument("--gzip-members", action="store_true", help="the gzip file as 32 concatenated members (round 2's measurement) instead of one")
    ap.add_argument("--gzip-modes", default="", help="comma list of extra gzip runs: t<N> (MDBG_HOST_GZIP_THREADS=N), nopool, branchy, round2 (both)")
    ap.add_argument("--no-bgzf", action="store_true")
    ap.add_argument("--reps", type=int, default=2, help="runs per thread count of the plain FASTA / FASTQ legs (the best is reported, all are listed)")
    ap.add_argument("--no-gzip", action="store_true", help="of the compressed legs, BGZF only")
    ap.add_argument("--bgzf-modes", default="", help="comma list of extra BGZF runs: copy (the round-2 path: MDBG_HOST_BGZF_COPY=1), t<N> (MDBG_HOST_BGZF_THREADS=N)")
    ap.add_argument("--batch-bases", default="", help="comma list: readSelection --batch-bases values to compare on the FASTA set (tool flag, not a reference flag)")
    a = ap.parse_args()
    from metamdbg_amd import capi, formats, synth
    work = tempfile.mkdtemp(prefix="mdbg_e2e_", dir=a.dir)
    res = {"host_threads": os.cpu_count(), "dir": a.dir}
    try:
        P = formats.Parameters(minimizer_size=15, kminmer_size=4, density=0.005, first_k=4, prev_k=4, hpc=True, data_type=0)
        threads = [int(t) for t in a.threads.split(",")]
        ctx = capi.Context(0)
        if a.reads:
            spec = synth.hifi_spec(a.reads, seed=42, read_len=10_000, coverage=50.0)
            fasta = os.path.join(work, "reads.fasta")
            res["fasta_write_s"] = write_reads(fasta, ctx, spec, a.reads, False)
        if a.fastq_reads:
            qspec = dataclasses.replace(synth.hifi_spec(a.fastq_reads, seed=42, read_len=10_000, coverage=50.0), with_quality=True)
            fastq = os.path.join(work, "reads.fastq")
            res["fastq_write_s"] = write_reads(fastq, ctx, qspec, a.fastq_reads, True)
        if a.ont_reads:
            ospec = synth.ont_spec(a.ont_reads, seed=43, read_len=20_000, coverage=50.0)
            ont_fastq = os.path.join(work, "ont.fastq")
            res["ont_write_s"] = write_reads(ont_fastq, ctx, ospec, a.ont_reads, True)
        ctx.close()

        def best_of(inputs, n_reads, label, reps=a.reps, extra_env=None):
            out = {}
            for t in threads:
                runs = [run_tool(os.path.join(work, "run"), inputs, t, P, extra_env=extra_env) for _ in range(reps)]
                b = min(runs, key=lambda r: r["total_s"])
                gbp = n_reads * 10_000 / 1e9
                b.update(gbp=gbp, gbps=gbp / b["total_s"], read_selection_gbps=gbp / b["read_selection_s"], all_total_s=[round(r["total_s"], 3) for r in runs],
                         all_read_selection_s=[round(r["read_selection_s"], 3) for r in runs])
                out[f"threads_{t}"] = b
                print(label, t, "threads: readSelection", sorted(b["all_read_selection_s"]), file=sys.stderr)
                print(label, t, "threads: %.2f s total (readSelection %.2f, graph %.2f) = %.1f Gbp/s" % (b["total_s"], b["read_selection_s"], b["graph_s"], b["gbps"]), file=sys.stderr, flush=True)
            return out
        if a.reads:
            res["fasta"] = best_of([fasta], a.reads, "fasta")
            for bb in [x for x in a.batch_bases.split(",") if x]:
                runs = [run_tool(os.path.join(work, "run"), [fasta], threads[0], P, rs_args=("--batch-bases", bb)) for _ in range(2)]
                b = min(runs, key=lambda r: r["total_s"])
                res[f"fasta_batch_bases_{bb}"] = b
                print("fasta --batch-bases", bb, "%.2f s total (readSelection %.2f)" % (b["total_s"], b["read_selection_s"]), file=sys.stderr, flush=True)
                for ln in b["trace_read_selection"]:
                    print("    ", ln, file=sys.stderr)
        if a.fastq_reads:
            res["fastq"] = best_of([fastq], a.fastq_reads, "fastq")
        if a.ont_reads:
            PO = formats.Parameters(minimizer_size=15, kminmer_size=4, density=0.005, first_k=4, prev_k=4, hpc=False, data_type=1, correction_density=0.025)
            out = {}
            for t in threads:
                runs = [run_tool(os.path.join(work, "run"), [ont_fastq], t, PO, rs_args=("--skip-correction",)) for _ in range(2)]
                b = min(runs, key=lambda r: r["total_s"])
                gbp = a.ont_reads * 20_000 / 1e9
                b.update(gbp=gbp, gbps=gbp / b["total_s"], read_selection_gbps=gbp / b["read_selection_s"], all_total_s=[round(r["total_s"], 3) for r in runs])
                out[f"threads_{t}"] = b
                print("ont", t, "threads: %.2f s total (readSelection %.2f, graph %.2f) = %.1f Gbp/s" % (b["total_s"], b["read_selection_s"], b["graph_s"], b["gbps"]), file=sys.stderr, flush=True)
            res["ont"] = out
        if a.gz_reads and a.reads:
            import multiprocessing as mp
            n = min(a.gz_reads, a.reads)
            # member borders at record starts: every record of r<index> is 10 002 + len(">r<index>") bytes; find them by scanning for "\n>"
            size_est = 0
            with open(fasta, "rb") as f:
                # byte offset of read n: walk in big strides (records are ~10 kB)
                pos, k = 0, 0
                bounds = [0]
                per = max(1, n // 32)
                while k < n:
                    step = min(per, n - k)
                    # records k .. k+step: sizes are 1 + len(str(i)) + 1 + 10000 + 1
                    pos += sum(len(str(i)) for i in range(k, k + step)) + step * 10_003
                    k += step
                    bounds.append(pos)
            if not a.no_gzip:
                t0 = time.perf_counter()
                gz = os.path.join(work, "reads.fasta.gz")
                if a.gzip_members:
                    with mp.Pool(min(32, os.cpu_count() or 1)) as pool:
                        parts = pool.map(_gz_member, [(fasta, bounds[i], bounds[i + 1]) for i in range(len(bounds) - 1)])
                    with open(gz, "wb") as f:
                        for p in parts:
                            f.write(p)
                    res["gzip_members"] = len(parts)
                else:
                    # ONE gzip member (what gzip / pigz write and sequencers deliver), compressed in pieces by all cores
                    import struct
                    with mp.Pool(min(32, os.cpu_count() or 1)) as pool:
                        parts = pool.map(_deflate_piece, [(fasta, bounds[i], bounds[i + 1], i == len(bounds) - 2) for i in range(len(bounds) - 1)])
                    crc, total = 0, 0
                    with open(gz, "wb") as f:
                        f.write(bytes.fromhex("1f8b08000000000004ff"))
                        for p, c, ln in parts:
                            f.write(p)
                            crc = _crc32_combine(crc, c, ln)
                            total += ln
                        f.write(struct.pack("<II", crc & 0xffffffff, total & 0xffffffff))
                    res["gzip_members"] = 1
                res["gzip_compress_s"] = time.perf_counter() - t0
                res["gzip_bytes"] = os.path.getsize(gz)
                del parts
                res["gzip"] = best_of([gz], n, "gzip", reps=2)
                for mode in [m for m in a.gzip_modes.split(",") if m]:
                    env = {"nopool": {"MDBG_HOST_GZIP_NO_POOL": "1"}, "branchy": {"MDBG_HOST_GZIP_BRANCHY": "1"},
                           "round2": {"MDBG_HOST_GZIP_NO_POOL": "1", "MDBG_HOST_GZIP_BRANCHY": "1"}}.get(mode) or {"MDBG_HOST_GZIP_THREADS": mode[1:]}
                    res[f"gzip_{mode}"] = best_of([gz], n, "gzip " + mode, reps=1, extra_env=env)
                os.unlink(gz)
            if not a.no_bgzf:
                # the same reads as BGZF (what samtools fastq / bam2fastq / bgzip write): every <= 64 KB block is inflated on its own
                t0 = time.perf_counter()
                with mp.Pool(min(32, os.cpu_count() or 1)) as pool:
                    parts = pool.map(_bgzf_part, [(fasta, bounds[i], bounds[i + 1]) for i in range(len(bounds) - 1)])
                bg = os.path.join(work, "reads.bgzf.fasta.gz")
                with open(bg, "wb") as f:
                    for p in parts:
                        f.write(p)
                    f.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))     # the BGZF end-of-file block
                res["bgzf_compress_s"] = time.perf_counter() - t0
                res["bgzf_bytes"] = os.path.getsize(bg)
                del parts
                res["bgzf"] = best_of([bg], n, "bgzf", reps=2)
                for mode in [m for m in a.bgzf_modes.split(",") if m]:
                    env = {"MDBG_HOST_BGZF_COPY": "1"} if mode == "copy" else {"MDBG_HOST_BGZF_THREADS": mode[1:]}
                    res[f"bgzf_{mode}"] = best_of([bg], n, "bgzf " + mode, reps=2, extra_env=env)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    line = json.dumps(res)
    if a.out:
        open(a.out, "w").write(line + "\n")
    print(line)


if __name__ == "__main__":
    main()
