#!/usr/bin/env python3
"""What `mdbg_tool asmStep` spends after its last output is closed (tools/e2e_steady.py: the tool's own clock says 0.46 s at 10 Gbp, its
parent's 0.59 s).  Runs the command over one FASTA file under a few environments and reports, per run, the parent's wall time, the tool's
last time line and their difference.

    python tools/exit_cost.py --reads 1000000 --out gpurun_out/exit_cost.json
"""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import e2e_steady  # noqa: E402


def one(work, fasta, P, threads, env_extra, cmd="asmStep"):
    tmp = os.path.join(work, "run", "tmp")
    shutil.rmtree(os.path.join(work, "run"), ignore_errors=True)
    for d in ("", "filter", "smallContigs", "checkpoints"):
        os.makedirs(os.path.join(tmp, d), exist_ok=True)
    P.save(os.path.join(tmp, "parameters.gz"))
    open(os.path.join(tmp, "input.txt"), "w").write(fasta + "\n")
    env = dict(os.environ, MDBG_TRACE="1")
    env.update(env_extra)
    argv = [e2e_steady.TOOL, cmd, tmp, tmp + "/read_data_init.txt", tmp + "/input.txt", "--threads", str(threads), "--min-read-quality", "0.000000"]
    if cmd == "asmStep":
        argv += ["--min-abundance", "0"]
    env.setdefault("MDBG_TOOL_EXIT_TRACE", "1")
    e0 = time.time()
    t0 = time.perf_counter()
    r = subprocess.run(argv, capture_output=True, text=True, env=env, timeout=240)
    wall = time.perf_counter() - t0
    e1 = time.time()
    assert r.returncode == 0, r.stderr[-1000:]
    marks = [(float(m.group(1)), m.group(2)) for m in re.finditer(r"\[mdbg_tool\]\s+([0-9.]+) s  (.*)", r.stderr)]
    last = marks[-1][0] if marks else None
    ep = re.search(r"clock started at ([0-9.]+), _exit at ([0-9.]+)", r.stderr)
    before = after = None
    if ep:
        before, after = round(float(ep.group(1)) - e0, 4), round(e1 - float(ep.group(2)), 4)
    return {"spawn_to_the_tools_clock_s": before, "exit_to_the_parents_wait_s": after, "wall_s": round(wall, 4), "tool_last_line_s": last, "outside_s": round(wall - last, 4) if last is not None else None,
            "last_lines": ["%.3f %s" % m for m in marks[-3:]]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--threads", type=int, default=32)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--dir", default="/dev/shm")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    from metamdbg_amd import capi, formats, synth
    work = tempfile.mkdtemp(prefix="mdbg_exit_", dir=a.dir)
    res = {}
    try:
        P = formats.Parameters(minimizer_size=15, kminmer_size=4, density=0.005, first_k=4, prev_k=4, hpc=True, data_type=0)
        ctx = capi.Context(0)
        fasta = os.path.join(work, "reads.fasta")
        e2e_steady.write_reads(fasta, ctx, synth.hifi_spec(a.reads, seed=42, read_len=10_000, coverage=50.0), a.reads, False)
        ctx.close()
        cases = [("as_shipped", {}), ("no_group_slabs_ahead", {"MDBG_TOOL_NO_GROUP_SLABS": "1"}), ("as_shipped_again", {}), ("no_group_slabs_ahead_again", {"MDBG_TOOL_NO_GROUP_SLABS": "1"})]
        cases = [(n, e, "asmStep") for n, e in cases]
        for name, env, cmd in cases:
            runs = [one(work, fasta, P, a.threads, env, cmd) for _ in range(a.reps)]
            res[name] = {"env": env, "command": cmd, "runs": runs, "best_wall_s": min(r["wall_s"] for r in runs), "median_outside_s": sorted(r["outside_s"] for r in runs)[len(runs) // 2],
                         "median_before_s": sorted(r["spawn_to_the_tools_clock_s"] for r in runs)[len(runs) // 2], "median_after_s": sorted(r["exit_to_the_parents_wait_s"] for r in runs)[len(runs) // 2]}
            print(name, [(r["wall_s"], r["tool_last_line_s"], r["exit_to_the_parents_wait_s"]) for r in runs], file=sys.stderr, flush=True)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        json.dump(res, open(a.out, "w"), indent=1)
    print(json.dumps({k: {"best_wall_s": v["best_wall_s"], "before": v["median_before_s"], "after": v["median_after_s"]} for k, v in res.items()}))


if __name__ == "__main__":
    main()
