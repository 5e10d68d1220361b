"""scan + purge once, then the multi-k loop k = 4 .. last over n x 10 kb HiFi reads (benchmark mode: reads only, the previous table is
the pass's own k - 1 output), twice, for rocprofv3 counter passes on the refined / index kernels (GPU box):
    python tools/index_once.py [n_reads] [last_k]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metamdbg_amd import capi, synth
ctx = capi.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
last = int(sys.argv[2]) if len(sys.argv) > 2 else 7
spec = synth.hifi_spec(n, seed=42, read_len=10000, coverage=50.0)
reads = ctx.reads_synthetic(spec)
corr = ctx.purge_palindromes(ctx.scan(reads, K=15, density=0.005, hpc=True), 4, 100)
reads.free()
for _ in range(2):
    prev = ctx.kminmer_count_first(corr, 4, 0)
    for k in range(5, last + 1):
        t = ctx.kminmer_count_refined(corr, None, k, prev) if k == 5 else ctx.kminmer_index(corr, None, k, prev)
        print(k, t.info(), t.stats(), flush=True)
        prev.free()
        prev = t
    prev.free()
