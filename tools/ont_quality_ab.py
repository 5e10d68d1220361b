"""FASTQ reads: the per-read quality sums in front of the scan (rounds 1 - 4) or beside it on the side stream (round 5), timed (GPU box):
n x 20 kb synthetic ONT reads with qualities resident in HBM, mdbg_scan (no HPC, l = 15, density 0.005) five times per setting, wall time of the
call and the HIP-event time of its kernels; the minimizers' order-independent digest must be the same.
    python tools/ont_quality_ab.py [n_reads]"""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metamdbg_amd import capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
ctx = capi.Context(0)
spec = synth.ont_spec(n, seed=43, read_len=20_000, coverage=50.0)
reads = ctx.reads_synthetic(spec)
out = {"reads": n, "bases": reads.info()["n_bases"]}
for name, opt in (("in_front_of_the_scan", 0), ("beside_the_scan", 1), ("in_front_of_the_scan_again", 0), ("beside_the_scan_again", 1)):
    ctx.set_option("scan_quality_stream", opt)
    walls, digest, ms = [], None, None
    for rep in range(5):
        ctx.synchronize()
        ctx.timing(True); ctx.timing_reset()
        t0 = time.perf_counter()
        m = ctx.scan(reads, K=15, density=0.005, hpc=False)
        ctx.synchronize()
        walls.append((time.perf_counter() - t0) * 1e3)
        ctx.timing(False)
        ms = {k: round(ctx.timing_get(k)[0], 3) for k in ("scan", "quality_sum", "scan_compact") if ctx.timing_get(k)[1]}
        if rep == 0:
            h = m.to_host()
            digest = (int(h["minimizers"].astype(np.uint64).sum()), int(h["qual"].astype(np.uint64).sum()), int(np.nan_to_num(h["mean_quality"]).astype(np.float64).sum() * 1000))
        m.free()
    out[name] = {"wall_ms_best": round(min(walls), 2), "wall_ms_all": [round(w, 2) for w in walls], "kernel_ms_last": ms, "digest": digest}
out["same_minimizers"] = len({tuple(out[k]["digest"]) for k in out if isinstance(out[k], dict)}) == 1
print(json.dumps(out, indent=1))
