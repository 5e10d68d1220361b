"""One rank of tests/test_gpu_multirank.py::test_two_ranks_multik_equal_one (launched by torch.distributed.run, gloo; every rank
on GPU 0): BASELINE.json configs[2] -- the multi-k loop k = 4 .. 8 in benchmark mode -- over reads sharded across the ranks with
metamdbg_amd/distributed.py; rank 0 gathers every k's records and writes them to <out>/k<k>.npy."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metamdbg_amd import capi, distributed as D, formats, synth  # noqa: E402

out, n_total, last_k = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
ctx = capi.Context(0)
spec = synth.hifi_spec(n_total, seed=77, read_len=6000, coverage=25.0)
per = n_total // world
reads = ctx.reads_synthetic(spec, first_read=rank * per, n_reads=per)
corr = ctx.purge_palindromes(ctx.scan(reads, K=15, density=0.005, hpc=True), 4, 100)
table = D.first_pass_sharded(ctx, corr, 4, 0)
for k in range(4, last_k + 1):
    if k > 4:
        table = D.next_k_sharded(ctx, corr, None, k, 4, records)
    records = D.allgather_records(table)
    if rank == 0:
        np.save(os.path.join(out, f"k{k}.npy"), np.frombuffer(records, formats.ABUNDANCE_DTYPE))
    table.free()
dist.barrier()
dist.destroy_process_group()
ctx.close()
