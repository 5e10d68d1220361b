"""scan + purge once, then the k = 4 first pass (count_insert_kernel and the kernels around it) three times over n x 10 kb HiFi
reads, for rocprofv3 counter passes on the table kernels (GPU box): python tools/insert_once.py [n_reads]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metamdbg_amd import capi, synth
ctx = capi.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
spec = synth.hifi_spec(n, seed=42, read_len=10000, coverage=50.0)
reads = ctx.reads_synthetic(spec)
corr = ctx.purge_palindromes(ctx.scan(reads, K=15, density=0.005, hpc=True), 4, 100)
reads.free()
for _ in range(3):
    t = ctx.kminmer_count_first(corr, 4, 0)
    print(t.info(), t.stats(), flush=True)
    t.free()
