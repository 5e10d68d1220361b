import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metamdbg_amd import capi, synth
ctx = capi.Context(0)
rng = np.random.default_rng(1)
for hpc in (False, True):
    for lens in ([100], [3000], [100, 50, 3000, 2048, 9000]):
        seqs = [bytes(synth.CODE2ASCII[rng.integers(0, 4, n)]) for n in lens]
        print("scan hpc", hpc, lens, flush=True)
        reads = ctx.reads_from_ascii(seqs)
        m = ctx.scan(reads, K=15, density=0.05, hpc=hpc)
        print(" ->", m.info(), flush=True)
