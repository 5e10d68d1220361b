"""What a "split by key, count in LDS" pass costs above firstK (GPU box; round-4 VERDICT item 3 asked for it measured, not argued).
The first pass's kernels ARE that pass -- radix multisplit of the instance records by key, one workgroup per bucket counting in LDS -- and
take any k: over n x 10 kb HiFi reads, scan + purge once, then at k = 5 .. last the loop's own pass (one-slot tables, index_tuning 3) and,
beside it, mdbg_kminmer_count_first at the same k over the same reads.  The partitioned pass gives the distinct keys of that k with their
counts; an index pass built on it would still have to ask the previous table twice per distinct key (what refine_slots_kernel does: about
1 ms) and write the rows -- so its time is a LOWER bound of such a pass.
    python tools/partitioned_count_by_k.py [n_reads] [last_k] > gpurun_out/.../partitioned_count_by_k.json"""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metamdbg_amd import capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
last = int(sys.argv[2]) if len(sys.argv) > 2 else 11
ctx = capi.Context(0)
spec = synth.hifi_spec(n, seed=42, read_len=10000, coverage=50.0)
reads = ctx.reads_synthetic(spec)
corr = ctx.purge_palindromes(ctx.scan(reads, K=15, density=0.005, hpc=True), 4, 100)
reads.free()
names = ("kminmer_split", "kminmer_prev_lookup", "kminmer_prev_image", "kminmer_insert", "kminmer_rescue", "kminmer_emit", "table_clear", "prefix_scan")


def timed(fn):
    best = None
    for _ in range(3):
        ctx.synchronize()
        ctx.timing(True); ctx.timing_reset()
        t0 = time.perf_counter()
        t = fn()
        ctx.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        ctx.timing(False)
        ms = {x: round(ctx.timing_get(x)[0], 3) for x in names if ctx.timing_get(x)[1]}
        r = {"wall_ms": round(wall, 3), "kernel_ms_total": round(sum(ms.values()), 3), "kernel_ms": ms, "records": t.info()["n_records"]}
        if best is None or r["wall_ms"] < best[0]["wall_ms"]:
            if best is not None:
                best[1].free()
            best = (r, t)
        else:
            t.free()
    return best


out = {"reads": n, "per_k": {}}
prev = ctx.kminmer_count_first(corr, 4, 0)
for k in range(5, last + 1):
    loop, t = timed(lambda: ctx.kminmer_count_refined(corr, None, k, prev) if k == 5 else ctx.kminmer_index(corr, None, k, prev))
    part, tp = timed(lambda: ctx.kminmer_count_first(corr, k, 0))
    # (the loop's table holds fewer keys than the count: a key of the loop also needs its two (k-1)-windows in the previous table)
    out["per_k"][str(k)] = {"the_loops_pass": loop, "partitioned_count_of_the_same_k": part}
    tp.free()
    prev.free()
    prev = t
    print(k, out["per_k"][str(k)], file=sys.stderr, flush=True)
prev.free()
print(json.dumps(out, indent=1))
