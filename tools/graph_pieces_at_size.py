#!/usr/bin/env python3
"""`mdbg_tool graph --firstpass` whole against the same pass in pieces (MDBG_TOOL_MAX_MINIMIZERS forced low) on a read set of real size:

    python tools/graph_pieces_at_size.py --reads 2000000 --pieces 4 --out gpurun_out/pieces.json

n x 10 kb synthetic HiFi reads as FASTA in /dev/shm -> readSelection once -> graph twice on copies of its files.  The two tables must be equal as
multisets (20-byte records and k-vectors) and the log lines the reference writes (solid / rescued counts, abundance checksum) identical.  Never
touches oracle/."""
from __future__ import annotations

import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
TOOL = os.path.join(ROOT, "metamdbg_amd", "bin", "mdbg_tool")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=2_000_000)
    ap.add_argument("--pieces", type=int, default=4)
    ap.add_argument("--threads", type=int, default=32)
    ap.add_argument("--dir", default="/dev/shm")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    from metamdbg_amd import capi, formats, synth
    import e2e_steady
    work = tempfile.mkdtemp(prefix="mdbg_pieces_", dir=a.dir)
    res = {"reads": a.reads, "pieces_asked": a.pieces}
    try:
        P = formats.Parameters(minimizer_size=15, kminmer_size=4, density=0.005, first_k=4, prev_k=4, hpc=True, data_type=0)
        ctx = capi.Context(0)
        fasta = os.path.join(work, "reads.fasta")
        e2e_steady.write_reads(fasta, ctx, synth.hifi_spec(a.reads, seed=42, read_len=10_000, coverage=50.0), a.reads, False)
        ctx.close()
        base = os.path.join(work, "whole")
        tmp = os.path.join(base, "tmp")
        for d in ("", "filter", "smallContigs", "checkpoints"):
            os.makedirs(os.path.join(tmp, d), exist_ok=True)
        P.save(os.path.join(tmp, "parameters.gz"))
        open(os.path.join(tmp, "input.txt"), "w").write(fasta + "\n")
        env = dict(os.environ, MDBG_TRACE="1")
        r = subprocess.run([TOOL, "readSelection", tmp, tmp + "/read_data_init.txt", tmp + "/input.txt", "--threads", str(a.threads), "--min-read-quality", "0.000000"],
                           capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        os.remove(fasta)
        n_min = (os.path.getsize(os.path.join(tmp, "read_data_corrected.txt")) - 5 * a.reads) // 4          # "u32 n; u8 flag; u32 m[n]" records
        res["minimizers"] = int(n_min)
        parts = os.path.join(work, "parts")
        shutil.copytree(base, parts)

        def graph(parent, extra_env):
            t0 = time.perf_counter()
            g = subprocess.run([TOOL, "graph", os.path.join(parent, "tmp"), "--threads", str(a.threads), "--min-abundance", "0", "--firstpass"],
                               capture_output=True, text=True, env=dict(env, **extra_env), timeout=600)
            dt = time.perf_counter() - t0
            assert g.returncode == 0, g.stderr[-2000:]
            log = open(os.path.join(parent, "metaMDBG.log")).read()
            lines = [ln.strip() for ln in log.splitlines() if "Nb solid" in ln or "Nb rescued" in ln or "Checksum kminmer abundance" in ln]
            pieces = [ln.strip() for ln in log.splitlines() if "The pass runs in" in ln]
            return dt, lines, pieces, [ln.strip() for ln in g.stderr.splitlines() if "[mdbg_tool]" in ln]
        res["whole_s"], whole_lines, _, res["whole_trace"] = graph(base, {})
        res["pieces_s"], part_lines, res["pieces_logged"], res["pieces_trace"] = graph(parts, {"MDBG_TOOL_MAX_MINIMIZERS": str(n_min // a.pieces + 1)})
        res["log_lines"] = whole_lines
        res["log_lines_equal"] = whole_lines == part_lines and len(whole_lines) == 3

        def table(parent):
            t = os.path.join(parent, "tmp")
            rec = formats.sorted_abundance_records(open(os.path.join(t, "kminmerData_abundance.txt"), "rb").read())
            vec = np.fromfile(os.path.join(t, "kminmerData_min.txt"), "<u4").reshape(-1, 4)
            return rec, vec[np.lexsort(vec.T[::-1])]
        (r0, v0), (r1, v1) = table(base), table(parts)
        res["records"] = int(len(r0))
        res["records_equal_as_multisets"] = bool(np.array_equal(r0, r1))
        res["vectors_equal_as_multisets"] = bool(np.array_equal(v0, v1))
        res["init_copy_equal"] = open(os.path.join(parts, "tmp", "kminmerData_abundance_init.txt"), "rb").read() == \
            open(os.path.join(parts, "tmp", "kminmerData_abundance.txt"), "rb").read()
    finally:
        shutil.rmtree(work, ignore_errors=True)
    ok = res.get("log_lines_equal") and res.get("records_equal_as_multisets") and res.get("vectors_equal_as_multisets") and res.get("init_copy_equal")
    res["all_equal"] = bool(ok)
    text = json.dumps(res, indent=1)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        open(a.out, "w").write(text)
    print(text)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
