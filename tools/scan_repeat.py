"""Back-to-back mdbg_scan launches over the bench batch, one context, nothing else on the device: does a launch stay at its first duration?
    python tools/scan_repeat.py [n_reads] [launches]
(Round 3: a scan launch takes 106 ms alone and 118-120 ms in the three-batches-in-flight bench whatever the other batches' kernels are given
-- fewer workgroups, fewer CUs; this separates "displaced by other kernels" from "the device under continuous load".)"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metamdbg_amd import capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ctx = capi.Context(0)
reads = ctx.reads_synthetic(synth.hifi_spec(n, seed=42, read_len=10000, coverage=50.0))
ctx.synchronize()
out = []
ctx.timing(True)
for i in range(reps):
    ctx.timing_reset()
    t0 = time.perf_counter()
    m = ctx.scan(reads, K=15, density=0.005, hpc=True)
    ctx.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    out.append((wall, ctx.timing_get("scan")[0]))
    m.free()
print("launch: wall ms / kernel ms (HIP events)")
for i, (w, k) in enumerate(out):
    print(f"{i:3d} {w:8.2f} {k:8.2f}")
