"""The drop-in as the reference calls it -- ONE `graph` PROCESS PER k, from files (pipeline/AssemblyPipeline.hpp:763-792,
graph/CreateMdbg.cpp:391-468) -- at BASELINE.json configs[2]'s size (GPU box): the corrected minimizer reads of n x 10 kb HiFi reads are
written to <scratch>/tmp/read_data_corrected.txt (1.55 GB at 10 M reads), the previous table of every k asked for is made by the library's
own loop (benchmark mode: reads only, empty unitig files) and written as kminmerData_abundance_prev.txt, then `mdbg_tool graph` runs as a
child process per k with MDBG_TRACE; per k: wall time of the process (the better of `reps`), the tool's own trace split into phases, and the
check that the table it wrote is the in-process one (record count, the checksum the tool logs).
    python tools/graph_per_k.py [n_reads] [k,k,...] [reps] > gpurun_out/graph_per_k.json
bench.py's leg `graph_per_k` calls run()."""
from __future__ import annotations

import json
import os
import re
import shutil
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metamdbg_amd import capi, formats, synth  # noqa: E402

TOOL = os.path.join(ROOT, "metamdbg_amd", "bin", "mdbg_tool")


def write_minimizer_reads_fast(path: str, mins: np.ndarray, offs: np.ndarray, chunk: int = 1 << 20) -> None:
    """read_data_corrected.txt: `u32 n; u8 circular = 0; u32 m[n]` per read (readSelection/ReadSelection.hpp:1420-1426), chunk by chunk."""
    offs = offs.astype(np.int64)
    with open(path, "wb") as f:
        for r0 in range(0, len(offs) - 1, chunk):
            r1 = min(r0 + chunk, len(offs) - 1)
            o = offs[r0: r1 + 1] - offs[r0]
            n = np.diff(o)
            vals = np.ascontiguousarray(mins[offs[r0]: offs[r1]], dtype="<u4")
            out = np.zeros(5 * (r1 - r0) + 4 * len(vals), dtype=np.uint8)
            start = 5 * np.arange(r1 - r0, dtype=np.int64) + 4 * o[:-1]                  # byte position of every record
            nb = n.astype("<u4").view(np.uint8).reshape(-1, 4)
            for b in range(4):
                out[start + b] = nb[:, b]
            # values: minimizer j of read r at start[r] + 5 + 4 j  ==  5 (r + 1) + 4 (o[r] + j)
            read_of = np.repeat(np.arange(r1 - r0, dtype=np.int64), n)
            pos = 5 * (read_of + 1) + 4 * np.arange(len(vals), dtype=np.int64)
            vb = vals.view(np.uint8).reshape(-1, 4)
            for b in range(4):
                out[pos + b] = vb[:, b]
            f.write(out.tobytes())


def parse_trace(stderr: str) -> list[tuple[float, str]]:
    return [(float(m.group(1)), m.group(2).strip()) for m in re.finditer(r"\[mdbg_tool\]\s+([0-9.]+) s\s+(.*)", stderr)]


def run(n_reads: int = 10_000_000, ks=(4, 5, 6, 11), reps: int = 2, scratch: str | None = None, ctx=None, threads: int = 32) -> dict:
    own = ctx is None
    if own:
        ctx = capi.Context(0)
    scratch = scratch or ("/dev/shm/mdbg_graph_per_k" if os.path.isdir("/dev/shm") else "/tmp/mdbg_graph_per_k")
    shutil.rmtree(scratch, ignore_errors=True)
    tmp = os.path.join(scratch, "tmp")
    for d in ("", "filter", "smallContigs", "checkpoints"):
        os.makedirs(os.path.join(tmp, d), exist_ok=True)
    out = {"reads": n_reads, "threads": threads, "per_k": {}}
    try:
        spec = synth.hifi_spec(n_reads, seed=42, read_len=10000, coverage=50.0)
        reads = ctx.reads_synthetic(spec)
        corr = ctx.purge_palindromes(ctx.scan(reads, K=15, density=0.005, hpc=True), 4, 100)
        reads.free()
        h = corr.to_host(full=False)
        t0 = time.perf_counter()
        write_minimizer_reads_fast(os.path.join(tmp, "read_data_corrected.txt"), h["minimizers"], h["offsets"])
        out["corrected_file_bytes"] = os.path.getsize(os.path.join(tmp, "read_data_corrected.txt"))
        out["corrected_file_write_s"] = round(time.perf_counter() - t0, 2)
        n_min = int(len(h["minimizers"]))
        del h
        # read_stats.txt as readSelection leaves it (only N50 / counts matter to `graph`)
        import struct
        open(os.path.join(tmp, "read_stats.txt"), "wb").write(struct.pack(formats.READ_STATS_STRUCT, n_reads, 10000, 0.00374, n_reads * 10000, 0.0, 10000, n_min))
        # the library's own loop: the expected table of every k asked for and the previous table of each
        expect, prev_bytes = {}, {}
        prev = ctx.kminmer_count_first(corr, 4, 0)
        expect[4] = (prev.info()["n_records"], int(prev.checksum()[0]))
        last = max(ks)
        for k in range(5, last + 1):
            if k in ks:
                prev_bytes[k] = prev.to_host()[0].tobytes()
            t = ctx.kminmer_count_refined(corr, None, k, prev) if k == 5 else ctx.kminmer_index(corr, None, k, prev)
            expect[k] = (t.info()["n_records"], int(t.checksum()[0]))
            prev.free()
            prev = t
        prev.free()
        corr.free()
        if own:
            ctx.close()
            ctx = None
        env = dict(os.environ, MDBG_TRACE="1")
        log_path = os.path.join(scratch, "metaMDBG.log")
        for k in ks:
            P = formats.Parameters(minimizer_size=15, kminmer_size=k, density=0.005, first_k=4, prev_k=max(4, k - 1), last_k=last, hpc=True, data_type=0)
            P.save(os.path.join(tmp, "parameters.gz"))
            if k > 4:
                open(os.path.join(tmp, "kminmerData_abundance_prev.txt"), "wb").write(prev_bytes[k])
                for name in ("unitig_data.txt", "unitigGraph_prev.nodes.bin", "unitigGraph.nodes.refined_abundances.bin"):
                    open(os.path.join(tmp, name), "wb").close()
            cmd = [TOOL, "graph", tmp, "--threads", str(threads)] + (["--min-abundance", "0", "--firstpass"] if k == 4 else [])
            best = None
            for _ in range(reps):
                if os.path.exists(log_path):
                    os.remove(log_path)
                t0 = time.perf_counter()
                r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
                wall = time.perf_counter() - t0
                if r.returncode != 0:
                    raise RuntimeError(f"mdbg_tool graph k={k} failed: {r.stderr[-800:]}")
                tr = parse_trace(r.stderr)
                if best is None or wall < best["wall_s"]:
                    best = {"wall_s": round(wall, 3), "trace": [[round(t, 3), w] for t, w in tr]}
            n_rec = os.path.getsize(os.path.join(tmp, "kminmerData_abundance.txt")) // 20
            m = re.search(r"Checksum kminmer abundance: (\d+)", open(log_path).read())
            best["records"] = n_rec
            best["table_equals_in_process_pass"] = bool(m) and (n_rec, int(m.group(1))) == expect[k]
            best["prev_table_bytes"] = len(prev_bytes.get(k, b""))
            # phases: differences between consecutive marks; after the last mark ("done") comes the exit
            ph, last_t = {}, 0.0
            for t, w in best["trace"]:
                ph[w] = round(t - last_t, 3)
                last_t = t
            ph["after the last mark (exit, wait)"] = round(best["wall_s"] - last_t, 3)
            best["phases_s"] = ph
            out["per_k"][str(k)] = best
    finally:
        shutil.rmtree(scratch, ignore_errors=True)
    out["all_tables_equal"] = all(v["table_equals_in_process_pass"] for v in out["per_k"].values())
    return out


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    ks = tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 else (4, 5, 6, 11)
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    print(json.dumps(run(n, ks, reps), indent=1))
