"""The scan of a batch in which a few reads carry an N (or soft-masked stretches) against the same batch without (GPU box):
    python tools/scan_with_n_time.py [n_reads] [one_in]
Both batches go through mdbg_reads_from_ascii; in the second, one read in `one_in` gets a few N and a lower-case stretch.  The block
kernel keeps the batch and only the marked reads take the general kernel (mdbg_scan: ScanArgs::skip); MDBG_SCAN_NO_FAST=1 shows what
the whole batch on the general kernel costs."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metamdbg_amd import capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
one_in = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
ctx = capi.Context(0)
spec = synth.hifi_spec(n, seed=42, read_len=10000, coverage=50.0)
src = ctx.reads_synthetic(spec)
bases, offs = src.export_ascii(0, n)
src.free()
L = 10000


def upload(arr):
    import ctypes as C
    h = C.c_void_p()
    ctx.check(capi.lib().mdbg_reads_from_ascii(ctx.h, capi._ptr(arr), None, capi._ptr(offs), n, C.byref(h)))
    return capi.Reads(ctx, h)


def timed(reads, label):
    best = None
    for it in range(4):
        ctx.timing(True); ctx.timing_reset()
        t0 = time.perf_counter()
        m = ctx.scan(reads, K=15, density=0.005, hpc=True)
        ctx.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        ctx.timing(False)
        ms, launches = ctx.timing_get("scan")
        nm = m.info()["n_minimizers"]
        m.free()
        if it and (best is None or wall < best[0]):
            best = (wall, ms, launches)
    print(f"{label:48s} reads={n} minimizers={nm}  wall {best[0]:7.2f} ms  scan kernels {best[1]:7.2f} ms in {best[2]} launches", flush=True)
    return nm


clean = upload(bases)
timed(clean, "clean batch")
clean.free()
marked = bases.copy()
rng = np.random.default_rng(1)
for r in range(0, n, one_in):
    a = r * L + int(rng.integers(100, L - 400))
    marked[a:a + 3] = ord("N")
    b = r * L + int(rng.integers(100, L - 400))
    marked[b:b + 200] |= 0x20            # a soft-masked stretch
dirty = upload(marked)
timed(dirty, f"one read in {one_in} with N / lower case")
dirty.free()
