"""Kernel time of mdbg_scan's plain-ACGT variants (GPU box): scan kernel alone, HIP events on the library's stream, best of 5 launches.
MDBG_SCAN_NO_BUMP=1: the same block-structured kernel writing into padded slots (+ compaction); MDBG_SCAN_NO_FAST=1: round 1's kernel."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metamdbg_amd import capi, synth
ctx = capi.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
spec = synth.hifi_spec(n, seed=42, read_len=10000, coverage=50.0)
reads = ctx.reads_synthetic(spec)
ctx.timing(True)
mode = "general" if os.environ.get("MDBG_SCAN_NO_FAST") else ("fast, padded slots" if os.environ.get("MDBG_SCAN_NO_BUMP") else "fast, rows at the cursor")
for hpc in (True, False):
    for filt in (True, False):
        best, comp = 1e9, 0.0
        for rep in range(5):
            ctx.timing_reset()
            m = ctx.scan(reads, K=15, density=0.005, hpc=hpc, apply_read_filters=filt)
            best = min(best, ctx.timing_get("scan")[0])
            comp = ctx.timing_get("scan_compact")[0]
            nm = m.info()["n_minimizers"]
            m.free()
        print(f"[{mode}] reads {n} hpc {hpc} filters {filt} minimizers {nm} scan ms {best:.3f} (+ compaction {comp:.3f})", flush=True)
