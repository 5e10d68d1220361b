"""HIP-event time of the k = 4 first pass's kernels over n x 10 kb HiFi reads, one context alone (GPU box):
    [MDBG_LIB=metamdbg_amd/libmdbg_hip_<tag>.so] python tools/insert_time.py [n_reads] [label]
With the INSERT_ABLATE builds (metamdbg_amd/build.py build_lib(tag=..., extra_flags=["-DINSERT_ABLATE=n"])) this is the breakdown of
count_insert_kernel: 3 = window hash only, 2 = + probe and claim (no count), 1 = + count (no per-instance slot store), 0 = all."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metamdbg_amd import capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
label = sys.argv[2] if len(sys.argv) > 2 else os.path.basename(capi.LIB_PATH)
ctx = capi.Context(0)
spec = synth.hifi_spec(n, seed=42, read_len=10000, coverage=50.0)
reads = ctx.reads_synthetic(spec)
corr = ctx.purge_palindromes(ctx.scan(reads, K=15, density=0.005, hpc=True), 4, 100)
reads.free()
names = ("kminmer_insert", "kminmer_rescue", "kminmer_emit", "table_clear", "prefix_scan")
best = None
for it in range(4):
    ctx.timing(True); ctx.timing_reset()
    t = ctx.kminmer_count_first(corr, 4, 0)
    ctx.synchronize()
    ctx.timing(False)
    ms = {k: round(ctx.timing_get(k)[0], 3) for k in names}
    st = t.stats()
    t.free()
    if it and (best is None or ms["kminmer_insert"] < best["kminmer_insert"]):
        best = ms
print(f"{label:28s} reads={n} instances={st['instances']} keys={st['keys']} slots={st['slots']}  " + " ".join(f"{k}={v}" for k, v in best.items()), flush=True)
