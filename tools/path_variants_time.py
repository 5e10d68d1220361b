"""Per-kernel timings of the variants bench.py does not cover (GPU box): FASTQ-style input (qualities), no HPC,
the ONT-like configuration, and the multi-k loop (k = 4..11) at the bench workload's size.

    python tools/path_variants_time.py [n_reads]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dataclasses

from metamdbg_amd import capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
ctx = capi.Context(0)
NAMES = ["scan", "scan_compact", "quality_sum", "complexity_exact", "purge_palindromes", "kminmer_insert", "kminmer_rescue", "kminmer_emit",
         "density_threshold", "kminmer_prev_lookup", "edge_index", "minimizer_census"]


def report(tag, t0):
    ctx.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    parts = {k: round(ctx.timing_get(k)[0], 3) for k in NAMES if ctx.timing_get(k)[1]}
    print(f"{tag:58s} wall {wall:8.2f} ms  {parts}", flush=True)


spec = synth.hifi_spec(n, seed=42, read_len=10000, coverage=50.0)
ctx.timing(True)
for with_q in (False, True):
    sp = dataclasses.replace(spec, with_quality=with_q)
    reads = ctx.reads_synthetic(sp)
    for hpc in (True, False):
        for rep in range(2):
            ctx.timing_reset(); t0 = time.perf_counter()
            m = ctx.scan(reads, K=15, density=0.005, hpc=hpc, min_read_quality=0.0)
            if rep: report(f"scan qual={with_q} hpc={hpc} minimizers={m.info()['n_minimizers']}", t0)
            if rep and hpc and not with_q: keep = m
            else: m.free()
    reads.free()

# multi-k loop on the HPC / no-quality minimizers (benchmark mode of SURVEY 8(d): previous table = own k-1 output)
corr = ctx.purge_palindromes(keep, 4, 100)
for rep in range(2):
    ctx.timing_reset(); t0 = time.perf_counter()
    t = ctx.kminmer_count_first(corr, 4, 0)
    if rep: report(f"k=4 first pass records={t.info()['n_records']}", t0)
ctx.timing_reset(); t0 = time.perf_counter()
e, ck = ctx.edge_index(t)
report(f"edge index of the k=4 nodes: {e.info()['n_records']} edges", t0)
e.free()
for k in range(5, 12):
    ctx.timing_reset(); t0 = time.perf_counter()
    t2 = ctx.kminmer_count_refined(corr, None, k, t) if k == 5 else ctx.kminmer_index(corr, None, k, t)
    report(f"k={k} {'refined' if k == 5 else 'index'} records={t2.info()['n_records']}", t0)
    t.free(); t = t2
# ONT-like: 20 kb reads, 2 % errors, qualities, no HPC, correction density + down-sampling
n_ont = max(n // 4, 1000)
ont = synth.SynthSpec(n_reads=n_ont, read_len=20000, seed=7, sub_rate=0.02, species_len=[int(n_ont * 20000 / 30)], species_weight=[1.0],
                      with_quality=True, name="ont")
if True:
    reads = ctx.reads_synthetic(ont)
    for rep in range(2):
        ctx.timing_reset(); t0 = time.perf_counter()
        m = ctx.scan(reads, K=13, density=0.025, hpc=False, apply_read_filters=False, quality_window=1)
        low = ctx.apply_density_threshold(m, 0.005)
        if rep: report(f"ONT-like correction scan + down-sampling: {m.info()['n_minimizers']} -> {low.info()['n_minimizers']}", t0)
