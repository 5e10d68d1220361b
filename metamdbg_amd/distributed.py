"""Exchange steps of the sharded first pass: one process per GPU, torch.distributed for the bytes
(backend "nccl" = RCCL over xGMI on the GPU box; "gloo" on CPU tensors in the tests).

Reads are sharded over the ranks; only the k-min-mer counts are global.  Keys are partitioned by
owner rank (include/mdbg_hip.h, mdbg_shard_*), so the merge is
    all-to-all   rows [hash_lo, hash_hi, count, packed vector...] -> owner     (exchange_by_owner)
    all-to-all   one u64 global count per row back to the sender             (reply_to_senders)
i.e. a reduce-scatter by key and its transpose.  xGMI is point-to-point, so an all-to-all keeps all
7 links of every GPU busy; a dense all-reduce would need a common key order first and nothing is
replicated here: no rank ever holds the global table.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def _staged(group) -> bool:
    """gloo has no all-to-all on device tensors: stage through the host (tests of the multi-rank logic on one GPU)."""
    return dist.get_backend(group) == "gloo"


def _all_to_all(out: torch.Tensor, inp: torch.Tensor, out_splits, in_splits, group) -> None:
    if inp.is_cuda and _staged(group):
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu().contiguous(), output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)
        out.copy_(o)
    else:
        dist.all_to_all_single(out, inp.contiguous(), output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)


def exchange_by_owner(rows: torch.Tensor, counts: list[int], group=None) -> tuple[torch.Tensor, list[int]]:
    """rows: (sum(counts), rw) grouped by destination rank.  Returns (rows this rank owns, rows received
    from each rank) -- the second list is the split of the reply."""
    world = dist.get_world_size(group)
    assert len(counts) == world and rows.shape[0] == sum(counts)
    send = torch.tensor(counts, dtype=torch.int64, device=rows.device)
    recv = torch.empty(world, dtype=torch.int64, device=rows.device)
    _all_to_all(recv, send, None, None, group)
    recv_counts = [int(x) for x in recv.tolist()]
    out = torch.empty((sum(recv_counts), rows.shape[1]), dtype=rows.dtype, device=rows.device)
    _all_to_all(out, rows, recv_counts, list(counts), group)
    return out, recv_counts


def reply_to_senders(reply: torch.Tensor, recv_counts: list[int], sent_counts: list[int], group=None) -> torch.Tensor:
    """reply: one value per received row (same order).  Returns one value per SENT row, in the order sent."""
    assert reply.shape[0] == sum(recv_counts)
    out = torch.empty((sum(sent_counts),) + tuple(reply.shape[1:]), dtype=reply.dtype, device=reply.device)
    _all_to_all(out, reply, list(sent_counts), list(recv_counts), group)
    return out
