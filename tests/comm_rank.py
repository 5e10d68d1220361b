"""One rank (one PROCESS) of the tests of the library's exchange through the C ABI alone (tests/test_gpu_multirank.py):
    python tests/comm_rank.py <rank> <n_ranks> <id file> <out .npy> <n_reads_total> [mode [share]]
Rank 0 writes the communicator id to the file (what a C++ pipeline without a communication layer would do: a file in the shared
tmp dir); every rank scans its contiguous share of one read set on its own GPU -- on GPU 0 with `share` = 1 -- and takes part in
mdbg_kminmer_count_first_sharded over the transport `mode` (rccl | peer | auto); the records of its share of the table go to <out>,
the transport it ended up with to <out>.mode."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metamdbg_amd import capi, synth  # noqa: E402

t_start = time.time()
rank, n_ranks, id_file, out, n_total = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], int(sys.argv[5])
mode = sys.argv[6] if len(sys.argv) > 6 else "auto"
share = len(sys.argv) > 7 and sys.argv[7] == "1"
ctx = capi.Context(0 if share else rank)
if rank == 0:
    uid = capi.Context.comm_unique_id()
    with open(id_file + ".tmp", "wb") as f:
        f.write(uid)
    os.replace(id_file + ".tmp", id_file)
else:
    t0 = time.time()
    while not os.path.exists(id_file):
        if time.time() - t0 > 60:
            sys.exit("no communicator id")
        time.sleep(0.05)
    uid = open(id_file, "rb").read()
comm = ctx.comm_create(uid, rank, n_ranks, mode)
with open(out + ".mode", "w") as f:
    f.write(comm.mode)
spec = synth.hifi_spec(n_total, seed=23, read_len=6000, coverage=25.0)
first, last = n_total * rank // n_ranks, n_total * (rank + 1) // n_ranks         # uneven shares when n_ranks does not divide
reads = ctx.reads_synthetic(spec, first_read=first, n_reads=last - first)
corr = ctx.purge_palindromes(ctx.scan(reads, K=15, density=0.005, hpc=True), 4, 100)
# (MDBG_TEST_DIE_BEFORE_PASS=<rank>:<pass>: that rank's process ends -- no destroy, no goodbye -- before its pass number <pass>)
die_rank, die_pass = (int(x) for x in os.environ.get("MDBG_TEST_DIE_BEFORE_PASS", "-1:-1").split(":"))
for n_pass in range(3):                  # three times: the communicator is reusable (and the words of its control block alternate)
    if rank == die_rank and n_pass == die_pass:
        os._exit(9)
    try:
        rec, vec = ctx.kminmer_count_first_sharded(comm, corr, 4, 0).to_host()
    except capi.MdbgError as exc:
        print(f"rank {rank}: pass {n_pass} failed after {time.time() - t_start:.1f} s: {exc}", file=sys.stderr, flush=True)
        os._exit(3)
np.save(out, rec)
comm.destroy()
ctx.close()
