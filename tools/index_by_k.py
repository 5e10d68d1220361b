"""The passes of the multi-k loop, pass by pass, to a deep k (GPU box): scan + purge once over n x 10 kb HiFi reads, then the loop
k = 4 .. last in benchmark mode (reads only, the previous table is the pass's own k - 1 output) for each "index_tuning" given, twice
(the better loop is kept); per k the HIP-event time of its kernels, the wall time of the call, instances, records, slots.  k = 5 .. 11 take
the window hash specialised per k, k >= 12 the generic one (csrc/kminmer_dev.hpp) -- the per-instance cost of the generic form is what
round-5 VERDICT item 1(c) asks for.  MDBG_TABLE_LOAD_PCT (environment) sets the load the tables are sized for.
    python tools/index_by_k.py [n_reads] [last_k] [tuning,tuning,...] > gpurun_out/index_by_k.json"""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metamdbg_amd import capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
last = int(sys.argv[2]) if len(sys.argv) > 2 else 11
tunings = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [-1]
ctx = capi.Context(0)
spec = synth.hifi_spec(n, seed=42, read_len=10000, coverage=50.0)
reads = ctx.reads_synthetic(spec)
corr = ctx.purge_palindromes(ctx.scan(reads, K=15, density=0.005, hpc=True), 4, 100)
reads.free()
names = ("kminmer_split", "kminmer_prev_lookup", "kminmer_prev_image", "kminmer_insert", "kminmer_rescue", "kminmer_emit", "table_clear", "prefix_scan")
out = {"reads": n, "last_k": last, "table_load_pct": os.environ.get("MDBG_TABLE_LOAD_PCT", "default"), "tunings": {}}
sums_ref = None
for tune in tunings:
    ctx.set_option("index_tuning", tune)
    best = None
    for rep in range(2):
        per_k, sums = {}, {}
        ctx.synchronize()
        t_loop = time.perf_counter()
        prev = ctx.kminmer_count_first(corr, 4, 0)
        for k in range(5, last + 1):
            ctx.synchronize()
            ctx.timing(True); ctx.timing_reset()
            t0 = time.perf_counter()
            t = ctx.kminmer_count_refined(corr, None, k, prev) if k == 5 else ctx.kminmer_index(corr, None, k, prev)
            ctx.synchronize()
            wall = (time.perf_counter() - t0) * 1e3
            ctx.timing(False)
            ms = {x: round(ctx.timing_get(x)[0], 3) for x in names if ctx.timing_get(x)[1]}
            st = t.stats()
            per_k[str(k)] = {"wall_ms": round(wall, 3), "kernel_ms_total": round(sum(ms.values()), 3), "kernel_ms": ms, "records": t.info()["n_records"],
                             "instances": st["instances"], "slots": st["slots"]}
            sums[str(k)] = [int(x) for x in t.checksum()]
            prev.free()
            prev = t
        prev.free()
        ctx.synchronize()
        loop_ms = (time.perf_counter() - t_loop) * 1e3
        if best is None or loop_ms < best["loop_ms_incl_first_pass"]:
            best = {"loop_ms_incl_first_pass": round(loop_ms, 2), "per_k": per_k}
        if sums_ref is None:
            sums_ref = sums
        best["tables_equal_to_first_tuning"] = sums == sums_ref
    out["tunings"][str(tune)] = best
print(json.dumps(out, indent=1))
