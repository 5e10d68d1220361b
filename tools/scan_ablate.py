"""What the pieces of scan_fast_kernel cost: one mdbg_scan of N x 10 kb HiFi reads (default 10 M), alone on the device, timed over five
launches after a warm-up, with the read filters (the complexity bound) on and off, with and without homopolymer compression.
GPU box: python tools/scan_ablate.py [n_reads]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metamdbg_amd import capi, synth
ctx = capi.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000000
spec = synth.hifi_spec(n, seed=42, read_len=10000, coverage=50.0)
reads = ctx.reads_synthetic(spec)
for hpc in (True, False):
    for filt in (True, False):
        ts = []
        for it in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m = ctx.scan(reads, K=15, density=0.005, hpc=hpc, apply_read_filters=filt)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
            nm = m.info()["n_minimizers"]
            m.free()
        print(f"hpc={int(hpc)} filters={int(filt)}  {nm} minimizers  best {min(ts[1:]) * 1e3:.2f} ms  median {sorted(ts[1:])[2] * 1e3:.2f} ms", flush=True)
