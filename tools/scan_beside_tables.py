"""What slows a scan launch beside another batch's table kernels: their instructions or their memory traffic?  (GPU box)
    python tools/scan_beside_tables.py
Context A scans the 10 M-read bench batch over and over; context B, on its own thread, keeps building first-pass tables
  (1) not at all                                   -- the scan alone
  (2) over the bench batch's 10 M reads            -- 344 M instances into a 4.3 GB table: 45.7 GB of random memory traffic per pass
  (3) over 300 000 reads, again and again          -- the same kernels and instruction mix, a 160 MB table that the 256 MB Infinity Cache holds
and reports the scan's launch time (HIP events) in each setting, with the table passes B completed meanwhile."""
import os
import sys
import threading
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from metamdbg_amd import capi, synth

a, b = capi.Context(0), capi.Context(0)
spec = synth.hifi_spec(10_000_000, seed=42, read_len=10000, coverage=50.0)
reads = a.reads_synthetic(spec)
corr_big = b.purge_palindromes(b.scan(reads, K=15, density=0.005, hpc=True), 4, 100)
small = b.reads_synthetic(synth.hifi_spec(300_000, seed=42, read_len=10000, coverage=50.0))
corr_small = b.purge_palindromes(b.scan(small, K=15, density=0.005, hpc=True), 4, 100)
b.set_option("table_blocks_per_cu", 1)


def run(label, corr):
    stop = threading.Event()
    done = [0, 0]

    def tables():
        while not stop.is_set():
            t = b.kminmer_count_first(corr, 4, 0)
            done[0] += 1
            done[1] += t.stats()["instances"]
            t.free()
    th = threading.Thread(target=tables) if corr is not None else None
    if th:
        b.kminmer_count_first(corr, 4, 0).free()          # table sized, pools warm
        th.start()
    times = []
    a.timing(True)
    t0 = time.perf_counter()
    for i in range(14):
        a.timing_reset()
        m = a.scan(reads, K=15, density=0.005, hpc=True)
        a.synchronize()
        times.append(a.timing_get("scan")[0])
        m.free()
    wall = time.perf_counter() - t0
    stop.set()
    if th:
        th.join()
    ts = sorted(times[2:])
    print(f"{label:58s} scan launch median {ts[len(ts) // 2]:7.2f} ms (min {ts[0]:.2f} max {ts[-1]:.2f});  B: {done[0]} table passes, "
          f"{done[1] / 1e6 / wall:8.1f} M instances/s", flush=True)


run("(1) scan alone", None)
run("(2) beside first passes over 10 M reads (4.3 GB table)", corr_big)
run("(3) beside first passes over 300 k reads (160 MB table)", corr_small)
