"""The passes above firstK in their forms, timed (GPU box): scan + purge once over n x 10 kb HiFi reads, then the loop k = 4 .. 11
(benchmark mode) with the one-slot tables -- the kernels of rounds 1 - 4 and round 5's (a slot's words in one trip, the insert's plain-load
first look, two windows of a lane in flight: mdbg_set_option "index_tuning"), each alone and together -- and with bucket tables (three keys
per 64-byte sector; the refined pass by look-ups: "index_table_form" / "refined_form"), each twice; per k the HIP-event time of its kernels and the wall time of the call, and
the tables' order-independent sums (they must be the same in both forms).
    python tools/index_forms_time.py [n_reads] [last_k] > gpurun_out/.../index_forms.json"""
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
out = {"reads": n, "forms": {}}
# (form, index_table_form, refined_form, index_tuning): tuning bit 0 a slot's words in one trip, bit 1 the insert's plain-load first look, bit 2 two windows in flight, bit 3 look-up and insert in one kernel
FORMS = (("slots_round4_kernels", 1, 1, 0), ("slots_wide", 1, 1, 1), ("slots_wide_fast", 1, 1, 3), ("slots_two_in_flight", 1, 1, 4), ("slots_all", 1, 1, 7),
         ("slots_fused", 1, 1, 11), ("slots_fused_without_the_first_look", 1, 1, 9),
         ("buckets", 0, 0, 7), ("buckets_refined_by_distinct_keys", 0, 1, 7))
for form, idx, ref, tune in FORMS:
    ctx.set_option("index_table_form", idx)
    ctx.set_option("refined_form", ref)
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
            per_k[str(k)] = {"wall_ms": round(wall, 3), "kernel_ms_total": round(sum(ms.values()), 3), "kernel_ms": ms, "records": t.info()["n_records"], "slots": t.stats()["slots"]}
            sums[str(k)] = [int(x) for x in t.checksum()]
            prev.free()
            prev = t
        prev.free()
        ctx.synchronize()
        loop_ms = (time.perf_counter() - t_loop) * 1e3
        if best is None or loop_ms < best["loop_ms_incl_first_pass"]:
            best = {"loop_ms_incl_first_pass": round(loop_ms, 2), "per_k": per_k, "sums": sums}
    out["forms"][form] = best
f = out["forms"]
out["tables_equal_in_all_forms"] = all(f[x]["sums"] == f["slots_all"]["sums"] for x in f)
for x in f:
    del f[x]["sums"]
print(json.dumps(out, indent=1))
