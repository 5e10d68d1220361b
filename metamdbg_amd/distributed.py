"""Exchange step of the multi-GPU first pass: one process per GPU, torch.distributed for the bytes
(backend "nccl" = RCCL over xGMI on the GPU box; "gloo" on CPU tensors in the tests).

Rows are int64 tensors of shape (n, row_words): [hash_lo, hash_hi, count, packed vector...]
(include/mdbg_hip.h).  Keys are partitioned by owner rank, so the merge is a reduce-scatter by
key (all-to-all + local reduce) followed by an all-gather of the owners' reduced slices -- together
an all-reduce of the count tables by key.  A dense ncclAllReduce would need the ranks to agree on
a common key order first and is bound by one xGMI link; the partitioned form keeps all 7 links busy.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def exchange_by_owner(rows: torch.Tensor, counts: list[int], group=None) -> torch.Tensor:
    """rows: (sum(counts), rw) grouped by destination rank.  Returns the rows this rank owns."""
    world = dist.get_world_size(group)
    assert len(counts) == world and rows.shape[0] == sum(counts)
    send = torch.tensor(counts, dtype=torch.int64, device=rows.device)
    recv = torch.empty(world, dtype=torch.int64, device=rows.device)
    dist.all_to_all_single(recv, send, group=group)
    recv_counts = [int(x) for x in recv.tolist()]
    out = torch.empty((sum(recv_counts), rows.shape[1]), dtype=rows.dtype, device=rows.device)
    dist.all_to_all_single(out, rows.contiguous(), output_split_sizes=recv_counts, input_split_sizes=list(counts), group=group)
    return out


def all_gather_rows(rows: torch.Tensor, group=None) -> torch.Tensor:
    """Concatenate every rank's (n_r, rw) rows, rank order."""
    world = dist.get_world_size(group)
    n = torch.tensor([rows.shape[0]], dtype=torch.int64, device=rows.device)
    sizes = [torch.empty(1, dtype=torch.int64, device=rows.device) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes) if sizes else 0
    pad = torch.zeros((mx, rows.shape[1]), dtype=rows.dtype, device=rows.device)
    pad[: rows.shape[0]] = rows
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)], dim=0)
