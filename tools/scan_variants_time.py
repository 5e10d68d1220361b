import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metamdbg_amd import capi, synth
ctx = capi.Context(0)
n = 1000000
spec = synth.hifi_spec(n, seed=42, read_len=10000, coverage=50.0)
reads = ctx.reads_synthetic(spec)
ctx.timing(True)
for hpc in (True, False):
    for filt in (True, False):
        ctx.timing_reset()
        m = ctx.scan(reads, K=15, density=0.005, hpc=hpc, apply_read_filters=filt)
        print("hpc", hpc, "filters", filt, m.info()["n_minimizers"], "scan ms %.3f" % ctx.timing_get("scan")[0], flush=True)
        m.free()
