"""One mdbg_scan of 1 M x 10 kb reads (HPC, filters on) after a warm-up, for rocprofv3 counter passes (GPU box)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metamdbg_amd import capi, synth
ctx = capi.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
spec = synth.hifi_spec(n, seed=42, read_len=10000, coverage=50.0)
reads = ctx.reads_synthetic(spec)
for _ in range(2):
    m = ctx.scan(reads, K=15, density=0.005, hpc=True, apply_read_filters=True)
    print(m.info()["n_minimizers"], flush=True)
    m.free()
