"""Which fraction of the k-window instances of a read set is NOT in the k-min-mer table of that k (a window whose previous abundances say "error":
min(prev[i], prev[i+1]) <= 1, graph/CreateMdbg.hpp:1440-1459) -- the instances for which an insert-first index pass would still have to ask the
previous table.  1 M HiFi reads, the windows of the first 3000 reads hashed by the oracle and looked up in the device's tables, k = 5 .. 24.
GPU box: python tools/index_miss_fraction.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metamdbg_amd import capi, synth
from oracle import pyoracle as orc
ctx = capi.Context(0)
spec = synth.hifi_spec(1_000_000, seed=42, read_len=10_000, coverage=50.0)
reads = ctx.reads_synthetic(spec)
mins = ctx.scan(reads, K=15, density=0.005, hpc=True)
reads.free()
corr = ctx.purge_palindromes(mins, 4, 100)
mins.free()
head = ctx.minimizers_slice(corr, 0, 3000).to_host(full=False)
m, off = head["minimizers"], head["offsets"].astype(np.int64)
prev = ctx.kminmer_count_first(corr, 4, 0)
for k in range(5, 25):
    t = ctx.kminmer_count_refined(corr, None, k, prev) if k == 5 else ctx.kminmer_index(corr, None, k, prev)
    lo, hi = [], []
    for r in range(3000):
        a, b = int(off[r]), int(off[r + 1])
        for i in range(a, b - k + 1):
            _, _, h_hi, h_lo = orc.kminmer_normalize_hash(m[i: i + k])
            lo.append(h_lo); hi.append(h_hi)
    got = t.lookup(np.array(lo, np.uint64), np.array(hi, np.uint64))
    print(f"k={k}: {len(lo)} instances of 3000 reads, {int((got == 0).sum())} not in the table = {float((got == 0).mean()):.3f}", flush=True)
    prev.free(); prev = t
