"""HIP-event time of the k = 4 first pass over n x 10 kb HiFi reads, one context alone, the one-table path against the partitioned
one (csrc/partition.hip) under several plans (GPU box):
    python tools/partition_time.py [n_reads] [plan ...]       plan = name:opt=value,opt=value   (options of mdbg_set_option)
Prints one JSON line per plan: the best of three runs (by the sum of the pass's kernels), counts, mdbg_first_pass_info."""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metamdbg_amd import capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
plans = sys.argv[2:] or ["one_table:first_pass_mode=1", "partitioned:first_pass_mode=2", "partitioned_2048:first_pass_mode=2,partition_lds_slots=2048"]
ctx = capi.Context(0)
ctx.set_option("pool_cache_percent", 90)
spec = synth.hifi_spec(n, seed=42, read_len=10000, coverage=50.0)
reads = ctx.reads_synthetic(spec)
corr = ctx.purge_palindromes(ctx.scan(reads, K=15, density=0.005, hpc=True), 4, 100)
reads.free()
names = ("kminmer_split", "kminmer_insert", "kminmer_rescue", "kminmer_emit", "table_clear", "prefix_scan")
ref = None
for plan in plans:
    label, _, opts = plan.partition(":")
    opts = dict((o.split("=")[0], int(o.split("=")[1])) for o in opts.split(",") if o)
    for name in ("first_pass_mode", "partition_bits", "partition_lds_slots", "partition_max_records"):
        ctx.set_option(name, opts.get(name, 0))
    best = None
    for it in range(4):
        ctx.timing(True); ctx.timing_reset()
        t0 = time.perf_counter()
        t = ctx.kminmer_count_first(corr, 4, 0)
        ctx.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        ctx.timing(False)
        ms = {k: round(ctx.timing_get(k)[0], 3) for k in names}
        ms["kernels"] = round(sum(ms.values()), 3)
        ms["wall"] = round(wall, 3)
        st, info, sums, fp = t.stats(), t.info(), t.checksum(), ctx.first_pass_info()
        t.free()
        if it and (best is None or ms["kernels"] < best["kernels"]):
            best = ms
    if ref is None:
        ref = (info, sums, st["keys"], st["instances"])
    print(json.dumps(dict(plan=label, reads=n, **best, stats=st, info=info, first_pass=fp,
                          equal_to_first_plan=(info, sums, st["keys"], st["instances"]) == ref)), flush=True)
