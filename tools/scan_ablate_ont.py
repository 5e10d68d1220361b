"""What the ONT variant of the scan pays for: N x 20 kb ONT reads with qualities (default 2 M), alone on the device, the scan kernel's own
time (mdbg_timing "scan") over three launches after a warm-up: as the ont leg runs it (qualities, repetitive list from the census),
without the list, with the qualities ignored, and both.  GPU box: python tools/scan_ablate_ont.py [n_reads]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metamdbg_amd import capi, synth
ctx = capi.Context(0)
ctx.timing(True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
spec = synth.ont_spec(n, seed=43, read_len=20_000, coverage=50.0)
reads = ctx.reads_synthetic(spec)
pre = ctx.scan(reads, K=15, density=0.025, hpc=False, apply_read_filters=False, ignore_qualities=True)
rep = ctx.repetitive_minimizers(pre)
pre.free()
print(f"{n} reads, repetitive list of {len(rep)}", flush=True)
for use_rep in (True, False):
    for ignore_q in (False, True):
        for filt in (True, False):
            best = None
            for it in range(4):
                ctx.timing_reset()
                m = ctx.scan(reads, K=15, density=0.005, hpc=False, repetitive=rep if use_rep else None, ignore_qualities=ignore_q, apply_read_filters=filt)
                torch.cuda.synchronize()
                ms = ctx.timing_get("scan")[0]
                q = ctx.timing_get("quality_sum")[0]
                nm = m.info()["n_minimizers"]
                m.free()
                if it and (best is None or ms < best[0]): best = (ms, q)
            print(f"repetitive={int(use_rep)} qualities={int(not ignore_q)} filters={int(filt)}  {nm} minimizers  scan kernel {best[0]:.2f} ms  quality_sum {best[1]:.2f} ms", flush=True)
# the qualities' mean (quality_sum_kernel) after the scan instead of beside it: what the scan kernel costs without that neighbour
ctx.set_option("scan_quality_stream", 0)
best = None
for it in range(4):
    ctx.timing_reset()
    m = ctx.scan(reads, K=15, density=0.005, hpc=False, repetitive=rep)
    torch.cuda.synchronize()
    ms, q = ctx.timing_get("scan")[0], ctx.timing_get("quality_sum")[0]
    m.free()
    if it and (best is None or ms < best[0]): best = (ms, q)
print(f"repetitive=1 qualities=1 filters=1, quality_sum AFTER the scan: scan kernel {best[0]:.2f} ms  quality_sum {best[1]:.2f} ms", flush=True)
