"""One rank of the two-GPU smoke test of the library's RCCL exchange (tests/test_gpu_multirank.py):
    python tests/rccl_rank.py <rank> <n_ranks> <id file> <out .npy> <n_reads_total>
Rank 0 writes the communicator id to the file; every rank scans its contiguous share of one read set on its own GPU and
takes part in mdbg_kminmer_count_first_sharded; the records of its share of the table go to <out>."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metamdbg_amd import capi, synth  # noqa: E402

rank, n_ranks, id_file, out, n_total = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], int(sys.argv[5])
ctx = capi.Context(rank)
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
comm = ctx.comm_create(uid, rank, n_ranks)
spec = synth.hifi_spec(n_total, seed=23, read_len=6000, coverage=25.0)
per = n_total // n_ranks
reads = ctx.reads_synthetic(spec, first_read=rank * per, n_reads=per)
corr = ctx.purge_palindromes(ctx.scan(reads, K=15, density=0.005, hpc=True), 4, 100)
for _ in range(2):                       # twice: the communicator is reusable
    rec, vec = ctx.kminmer_count_first_sharded(comm, corr, 4, 0).to_host()
np.save(out, rec)
comm.destroy()
ctx.close()
