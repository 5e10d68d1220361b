#!/usr/bin/env python3
"""bench.py -- Gbp/s through the minimizer + k-min-mer step on MI355X (BASELINE.json metric).

A step = one pass of the hot path over one batch of synthetic HiFi reads already resident in HBM
(2-bit packed): reads -> minimizers (HPC, l=15, density 0.005) -> palindrome purge -> k-min-mer
table at k=4 (count + rescue).  N=1 workload = BASELINE.json configs[1]: 1 M x 10 kb reads.
Three batches are in flight per GPU (--in-flight): consecutive steps run on their own library contexts
(own HIP stream, memory pool and host thread each), so the atomic-bound table kernels and the exchanges of
one batch overlap the ALU-bound scan of another; every step is still a complete pass over its batch.
N>1: one process per GPU, every rank owns its own shard of the same size (weak scaling); only the
k-min-mer counts are global: rows go to their owner rank and the global counts come back, two
all-to-alls over RCCL (metamdbg_amd/distributed.py, include/mdbg_hip.h mdbg_shard_*).

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP-event timed on the library's
stream) and `cpu_baseline` (the reference's own code, oracle/_ref/refdrv, timed on this box's cores
on a bounded sample of the same reads).
"""
from __future__ import annotations

import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# HIP multiplexes streams onto a few hardware queues (4 by default).  With 4, the streams of the two library contexts
# ended up on one queue in every run under torch.distributed + RCCL (no overlap of the batches in flight); with 8 they
# rarely do, and main() checks and repairs the rest.  Must be set before HIP starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
K_MINIMIZER, DENSITY, KMINMER = 15, 0.005, 4


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads", type=int, default=1_000_000, help="reads per GPU per step (10 kb each)")
    ap.add_argument("--read-len", type=int, default=10_000)
    ap.add_argument("--in-flight", type=int, default=3, help="batches processed concurrently per GPU (own context, stream and host thread each)")
    ap.add_argument("--cpu-sample", type=int, default=200_000, help="reads in the CPU-baseline sample (0 = skip)")
    return ap.parse_args()


def cpu_baseline(ctx, reads, n_sample: int) -> dict | None:
    """Time the REFERENCE's readSelection + graph --firstpass (oracle/_ref/refdrv) on the first
    n_sample reads of the batch, all host cores."""
    from metamdbg_amd import formats
    refdrv = os.path.join(ROOT, "oracle", "_ref", "refdrv")
    if n_sample <= 0 or not os.path.exists(refdrv):
        return None
    # the reference's thread scaling collapses past a few dozen threads (its graph command did not finish
    # in 60 s with 256 threads on a 0.2 Gbp sample, 0.8 s with 8): use what its README / test scripts use
    cores = min(os.cpu_count() or 1, 32)
    work = tempfile.mkdtemp(prefix="mdbg_cpu_")
    try:
        bases, offs = reads.export_ascii(0, n_sample)
        fasta = os.path.join(work, "sample.fasta")
        with open(fasta, "wb") as f:
            for r in range(n_sample):
                f.write(b">r%d\n" % r)
                f.write(bases[int(offs[r]): int(offs[r + 1])].tobytes())
                f.write(b"\n")
        tmp = os.path.join(work, "tmp")
        for d in ("", "filter", "smallContigs", "checkpoints"):
            os.makedirs(os.path.join(tmp, d), exist_ok=True)
        formats.Parameters(minimizer_size=K_MINIMIZER, kminmer_size=KMINMER, density=DENSITY, first_k=4, prev_k=4,
                           hpc=True, data_type=0).save(os.path.join(tmp, "parameters.gz"))
        with open(os.path.join(tmp, "input.txt"), "w") as f:
            f.write(fasta + "\n")
        t0 = time.perf_counter()
        subprocess.run([refdrv, "readSelection", tmp, os.path.join(tmp, "read_data_init.txt"), os.path.join(tmp, "input.txt"),
                        "--threads", str(cores), "--min-read-quality", "0.000000"], check=True, timeout=300,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        t1 = time.perf_counter()
        subprocess.run([refdrv, "graph", tmp, "--threads", str(cores), "--min-abundance", "0", "--firstpass"], check=True, timeout=300,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        t2 = time.perf_counter()
        nbases = int(offs[n_sample])
        return {"value": nbases / 1e9 / (t2 - t0), "unit": "Gbp/s", "cores": cores, "kind": "reference",
                "sample": f"first {n_sample} reads ({nbases / 1e9:.2f} Gbp) of the batch as FASTA on local disk: "
                          f"readSelection {t1 - t0:.2f} s + graph --firstpass {t2 - t1:.2f} s "
                          f"(the reference's graph command also builds the graph after the table), --threads {cores} "
                          f"of {os.cpu_count()} hardware threads",
                "read_selection_gbps": nbases / 1e9 / (t1 - t0)}
    except Exception as exc:  # the baseline is reported, never required
        return {"value": None, "unit": "Gbp/s", "cores": cores, "kind": "reference", "sample": f"failed: {exc}"}
    finally:
        shutil.rmtree(work, ignore_errors=True)


def measured_traffic(reads: int, read_len: int):
    """HBM bytes per scan launch from the committed rocprofv3 PMC passes (profiles/rNN_scan_traffic.json),
    valid only for the workload they were collected on; None otherwise."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_scan_traffic.json"))):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get("reads") == reads and d.get("read_len") == read_len:
            best = d
    return None if best is None else best["traffic_bytes_per_launch"]


def main() -> None:
    args = parse_args()
    # The one JSON line must be the only thing on stdout: RCCL prints a version banner through C stdio, which a redirected
    # stdout delivers at exit -- after the line.  Everything else that goes to file descriptor 1 is sent to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import numpy as np
    import torch
    from metamdbg_amd import capi, synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    if os.environ.get("MDBG_BENCH_SHARE_GPU") == "1":      # test hook: every rank on device 0 (multi-rank logic on a 1-GPU box)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    force_exchange = os.environ.get("MDBG_BENCH_FORCE_EXCHANGE") == "1"    # exercise the sharded path on one GPU
    if world > 1 or force_exchange:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        backend = os.environ.get("MDBG_BENCH_BACKEND", "nccl")   # "gloo": exchanges staged through the host (test hook)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    # IN_FLIGHT batches are processed concurrently, each by its own host thread on its own library context (own HIP
    # stream and memory pool) over its own copy of the reads: the k-min-mer kernels of one batch (bound by the atomic
    # rate) and the gaps between launches overlap with the scan of the other (bound by the vector ALU).  Step i runs
    # on slot i % IN_FLIGHT; every step is still the complete pass over one batch.
    n_slots = max(1, args.in_flight)
    if n_slots > 1:
        # keep the table kernels' footprint small beside the other batches' scans: 1 / 2 / 3 / 4 resident blocks per CU give
        # 676 / 672 / 657 / 643 Gbp/s on one GPU and 489 / 479 / 463 on the per-rank workload of an 8-GPU job run through
        # the sharded path (profiles/r01g_table_footprint_sweep.txt)
        os.environ.setdefault("MDBG_TABLE_BLOCKS_PER_CU", "1")
    slots = []
    # one metagenome for the job (MDBG_BENCH_SPEC_RANKS: test hook, the per-rank workload of an N-rank job on one GPU)
    spec_ranks = int(os.environ.get("MDBG_BENCH_SPEC_RANKS", world))
    spec = synth.hifi_spec(args.reads * spec_ranks, seed=42, read_len=args.read_len, coverage=50.0)
    for _ in range(n_slots):
        c = capi.Context(local_rank)
        slots.append((c, c.reads_synthetic(spec, first_read=rank * args.reads, n_reads=args.reads)))   # rank r owns reads [r*n, (r+1)*n)
    ctx, reads = slots[0]
    info = ctx.device_info()
    n_bases = reads.info()["n_bases"]
    rw = capi.lib().mdbg_row_words(KMINMER)
    exchange = world > 1 or force_exchange

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    trace = os.environ.get("MDBG_BENCH_TRACE") == "1"
    phases: dict = {}
    # the exchanges of the sharded pass use ONE communicator: they run in global step order on every rank, one at a
    # time (a step's scan is several times longer than its exchange, so the turn-taking costs nothing)
    turn = threading.Condition()
    next_exchange = [0]

    def step(slot: int, index: int):
        ctx, reads = slots[slot]
        mins = ctx.scan(reads, K=K_MINIMIZER, density=DENSITY, hpc=True)
        corr = ctx.purge_palindromes(mins, 4, 100)
        if not exchange:
            table = ctx.kminmer_count_first(corr, KMINMER, 0)
        else:
            from metamdbg_amd import distributed as D
            tr = [time.perf_counter()] if trace else None
            def mark(name):
                if tr is not None:
                    ctx.synchronize()
                    tr.append(time.perf_counter())
                    phases[name] = phases.get(name, 0.0) + (tr[-1] - tr[-2]) * 1e3
            sh = ctx.shard_begin(corr, KMINMER, world)
            mark("begin")
            sent = [int(c) for c in sh.counts]
            with turn:
                turn.wait_for(lambda: next_exchange[0] >= index)
            try:
                send = torch.as_tensor(capi.DeviceView(sh.d_rows, (sh.n_rows, rw)), device="cuda") if sh.n_rows else \
                    torch.empty((0, rw), dtype=torch.int64, device="cuda")
                mine, got = D.exchange_by_owner(send, sent)
                torch.cuda.current_stream().synchronize()      # not the device: the other slot keeps running
                mark("all_to_all_rows")
                d_reply = sh.reduce(mine.data_ptr(), mine.shape[0])
                reply = torch.as_tensor(capi.DeviceView(d_reply, (mine.shape[0],)), device="cuda") if mine.shape[0] else \
                    torch.empty((0,), dtype=torch.int64, device="cuda")
                mark("reduce")
                glob = D.reply_to_senders(reply, got, sent)
                torch.cuda.current_stream().synchronize()
                mark("all_to_all_reply")
            finally:
                with turn:
                    next_exchange[0] = index + 1
                    turn.notify_all()
            table = sh.finish(glob.data_ptr(), 0)
            sh.free()
            mark("finish")
        n_min = mins.info()["n_minimizers"]
        ti = table.info()
        for o in (table, corr, mins):
            o.free()
        return n_min, ti

    results: dict = {}
    errors: list = []

    def run_steps(slot: int, indices: list):
        try:
            torch.cuda.set_device(local_rank)                  # the current device is per thread
            for i in indices:
                results[i] = step(slot, i)
            slots[slot][0].synchronize()
        except BaseException as exc:                           # surface it in the main thread
            errors.append(exc)
            with turn:
                next_exchange[0] = 1 << 60                     # never block the other slot on a dead one
                turn.notify_all()

    def run_phase(first: int, count: int):
        """Steps first .. first+count-1, step i on slot i % n_slots, the slots concurrently."""
        work = [[i for i in range(first, first + count) if i % n_slots == sl] for sl in range(n_slots)]
        threads = [threading.Thread(target=run_steps, args=(sl, w)) for sl, w in enumerate(work) if w]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]

    # warm-up: at least one step per slot (pools, table sizing, RCCL set-up), in whole rounds so that the timed steps
    # start on slot 0
    n_warm = max(args.warmup, n_slots)
    n_warm += (-n_warm) % n_slots
    run_phase(0, n_warm)

    # HIP multiplexes streams onto a few hardware queues; when the streams of two contexts land on the same queue their
    # kernels run one after the other and the batches in flight do not overlap at all (GPU_MAX_HW_QUEUES=1 reproduces
    # it: 520 instead of 615 Gbp/s).  Probe it with purely local steps (no collective, every rank decides alone): two
    # concurrent steps must take clearly less than twice one step; if not, give slot 1 a new context -- a new stream --
    # and look again.
    def local_step(slot: int):
        c, r = slots[slot]
        mins = c.scan(r, K=K_MINIMIZER, density=DENSITY, hpc=True)
        corr = c.purge_palindromes(mins, 4, 100)
        t = c.kminmer_count_first(corr, KMINMER, 0)
        for o in (t, corr, mins):
            o.free()
        c.synchronize()

    probe = None
    if n_slots > 1 and args.steps > 1:
        def timed(fn):
            t = time.perf_counter()
            fn()
            return time.perf_counter() - t

        def both():
            th = [threading.Thread(target=lambda sl=sl: (torch.cuda.set_device(local_rank), local_step(sl), local_step(sl))) for sl in (0, 1)]
            for t in th:
                t.start()
            for t in th:
                t.join()

        for attempt in range(4):
            local_step(0); local_step(1)                       # pools of the local path
            one = min(timed(lambda: local_step(0)) for _ in range(2))
            two = timed(both) / 2.0                            # per concurrent pair of steps
            probe = {"one_step_ms": one * 1e3, "two_concurrent_steps_ms": two * 1e3, "ratio": two / one, "contexts_replaced": attempt}
            if two < 1.85 * one:
                break
            c, r = slots[1]
            r.free(); c.close()
            c = capi.Context(local_rank)
            slots[1] = (c, c.reads_synthetic(spec, first_read=rank * args.reads, n_reads=args.reads))
    for c, _ in slots:
        c.timing(True)
        c.timing_reset()
    barrier()
    t0 = time.perf_counter()
    run_phase(n_warm, args.steps)
    barrier()
    dt = time.perf_counter() - t0
    for c, _ in slots:
        c.timing(False)
    n_min, ti = results[n_warm + args.steps - 1]
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    totals = torch.tensor([ti["n_records"], ti["n_solid"]], dtype=torch.int64, device="cuda")   # last step, summed over ranks
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(totals, op=dist.ReduceOp.SUM)
    dt = float(tmax.item())

    def timing_get(name):
        tot = [c.timing_get(name) for c, _ in slots]
        return sum(t[0] for t in tot), sum(t[1] for t in tot)

    names = ["scan", "scan_compact", "purge_palindromes", "kminmer_insert", "kminmer_rescue", "kminmer_emit"]
    if exchange:
        names += ["shard_rows", "shard_reduce"]
    ktimes = {k: timing_get(k) for k in names}
    scan_ms, scan_n = ktimes["scan"]
    scan_avg_s = scan_ms / 1e3 / max(scan_n, 1)
    # algorithmic bytes of one scan launch: 0.25 B per base read + 10 B per emitted minimizer
    alg_bytes = 0.25 * n_bases + 10.0 * n_min
    achieved = alg_bytes / scan_avg_s / 1e9 if scan_avg_s > 0 else 0.0
    # compute floor of the reference's algorithm on this part: one Murmur3 per homopolymer-compressed position
    hpc_positions = int(0.75 * n_bases)            # HPC keeps 3/4 of uniform random bases
    clock_hz = 2.4e9
    hash_floor_ms = hpc_positions / 64 * 186 / (info["n_cu"] * 4) / clock_hz * 1e3

    if rank == 0:
        base = cpu_baseline(ctx, reads, min(args.cpu_sample, args.reads)) if world == 1 else None
        total_bases = n_bases * world * args.steps
        out = {
            "metric": "Gbp/s through minimizer+k-min-mer step; bit-exact k-min-mer table vs ref",
            "value": total_bases / 1e9 / dt, "unit": "Gbp/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"{args.reads} synthetic HiFi reads x {args.read_len} bp per GPU (seed 42, 0.1% substitutions, "
                                   f"4 species, 50x), HPC on, l={K_MINIMIZER}, density {DENSITY}, single k iteration k={KMINMER} "
                                   "(count + rescue); inputs 2-bit packed and resident in HBM",
                       "reads_per_gpu": args.reads, "read_len": args.read_len, "minimizers_per_step": int(n_min),
                       "kminmer_records": int(totals[0].item()), "solid": int(totals[1].item()),
                       "batches_in_flight": n_slots, "overlap_probe": probe, "device": info["arch"], "cus": info["n_cu"]},
            "roofline": {"bound": "hbm", "kernel": "scan_kernel<HPC>", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(args.reads, args.read_len),
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": scan_avg_s * 1e3,
                         "concurrent_launches": n_slots,
                         "concurrency_note": (f"{n_slots} batches are in flight: a scan launch shares the device with the other batches' "
                                              "table kernels and lasts longer than alone (13.8 ms with --in-flight 1: 209 GB/s, 2.6 % of "
                                              "peak, 64 % of the hash floor); scans of different batches never overlap each other")
                                             if n_slots > 1 else None,
                         "note": "integer-hash kernel: Murmur3_x64_128 of every HPC position (17 integer multiplies, half-rate VALU) "
                                 "puts the ceiling at the VALU, far below HBM (DESIGN.md 4.1)",
                         # the hash alone, measured in isolation at full occupancy (tools/ubench/hash_rates.hip,
                         # profiles/r01_hash_rates_gfx950.txt): 186 cycles per 64 hashes per SIMD
                         "valu_floor": {"hash_cycles_per_64": 186, "hpc_positions_per_launch": hpc_positions,
                                        "floor_ms": hash_floor_ms, "frac": hash_floor_ms / (scan_avg_s * 1e3) if scan_avg_s > 0 else None}},
            "kernel_ms_per_step": {k: v[0] / args.steps for k, v in ktimes.items()},
            "cpu_baseline": base,
        }
        if base and base.get("value"):
            out["speedup_vs_cpu_reference"] = out["value"] / base["value"]
        if trace and phases:
            out["exchange_phase_ms_per_step_incl_warmup"] = {k: v / (args.steps + n_warm) for k, v in phases.items()}
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    for c, r in slots:
        r.free()
        c.close()


if __name__ == "__main__":
    main()
