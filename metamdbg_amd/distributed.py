"""Exchange steps of the sharded first pass moved by torch.distributed (a HARNESS path: tests, MDBG_BENCH_EXCHANGE=torch;
backend "nccl" = RCCL over xGMI on the GPU box; "gloo" on CPU tensors in the tests).  The product's exchange is inside the
library: mdbg_comm_create_mode / mdbg_shard_exchange (csrc/multigpu.hip: peer copies or RCCL).

Reads are sharded over the ranks; only the k-min-mer counts are global.  Keys are partitioned by
owner rank (include/mdbg_hip.h, mdbg_shard_*), so the merge is
    all-to-all   rows [hash_lo, hash_hi, count] (3 x u64: vectors never travel) -> owner   (exchange_by_owner)
    all-to-all   one u64 global count per row back to the sender             (reply_to_senders)
i.e. a reduce-scatter by key and its transpose.  xGMI is point-to-point, so an all-to-all keeps all
7 links of every GPU busy; a dense all-reduce would need a common key order first and nothing is
replicated here: no rank ever holds the global table.
"""
from __future__ import annotations

import contextlib
import threading

import torch
import torch.distributed as dist


def _staged(group) -> bool:
    """gloo has no all-to-all on device tensors: stage through the host (tests of the multi-rank logic on one GPU)."""
    return dist.get_backend(group) == "gloo"


def _all_to_all(out: torch.Tensor, inp: torch.Tensor, out_splits, in_splits, group) -> None:
    if dist.get_world_size(group) == 1:     # nothing to exchange (and RCCL's send/receive to self is not to be trusted with GBs)
        out.copy_(inp.reshape(out.shape))
        return
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


class ExchangeGate:
    """Several batches in flight on one GPU of an N > 1 job: an exchange over RCCL does not run beside a scan of another batch.

    RCCL's device kernel needs 37.6 KB of LDS a block (profiles/round4_g_rccl_device_kernel_resources_gfx950.txt); four blocks of the scan
    kernel leave 31 KB of a CU and the scan's grid refills every place a retiring block frees, so each RCCL launch of an exchange would wait
    for a whole scan to drain ("the exchange gate").  Host threads wrap their scan in ``with gate.scan():`` and their exchange in
    ``with gate.exchange():`` -- an exchange waits for the scans in flight to end and holds new ones back until it is over.  A disabled gate
    does nothing.  No ordering among exchanges is imposed here (the caller's turn-taking does that); several at once are allowed."""

    def __init__(self, enabled: bool = True):
        self.enabled = bool(enabled)
        self._cv = threading.Condition()
        self._scans = 0
        self._exchanges = 0
        self.waited_ms = 0.0                     # time exchanges spent waiting for scans to end (all threads)

    @contextlib.contextmanager
    def scan(self):
        if not self.enabled:
            yield
            return
        with self._cv:
            self._cv.wait_for(lambda: self._exchanges == 0)
            self._scans += 1
        try:
            yield
        finally:
            with self._cv:
                self._scans -= 1
                self._cv.notify_all()

    @contextlib.contextmanager
    def exchange(self):
        if not self.enabled:
            yield
            return
        import time
        t0 = time.perf_counter()
        entered = False
        try:
            with self._cv:
                self._exchanges += 1             # from here on no new scan starts
                entered = True
                try:
                    self._cv.wait_for(lambda: self._scans == 0)
                finally:
                    self.waited_ms += (time.perf_counter() - t0) * 1e3
            yield
        finally:                                 # also when the wait itself was interrupted: a count left up would hold every scan for good
            if entered:
                with self._cv:
                    self._exchanges -= 1
                    self._cv.notify_all()


class PeerFailure(RuntimeError):
    """Another rank could not go on (the harness's counterpart of MDBG_EPEER): nothing was exchanged in this phase."""

    def __init__(self, rank: int, code: int, phase: str):
        super().__init__(f"rank {rank} failed {phase} (code {code}); nothing was exchanged")
        self.rank, self.code, self.phase = rank, code, phase


def agree(local_code: int, phase: str, group=None, device=None) -> None:
    """Every rank says whether it got this far (0) or not (a negative code) and learns what the others said -- the protocol of
    mdbg_shard_exchange (include/mdbg_hip.h, "Failure behaviour of the collective calls"): a rank that failed locally still
    takes part in this small all-gather, so its peers raise PeerFailure instead of waiting in the next all-to-all for rows
    that never come.  The failing rank itself returns normally: its caller re-raises the local error."""
    world = dist.get_world_size(group)
    dev = device if device is not None and not _staged(group) else "cpu"
    mine = torch.tensor([int(local_code)], dtype=torch.int64, device=dev)
    everyone = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(everyone, mine, group=group)
    if local_code == 0:
        for r, t in enumerate(everyone):
            if int(t.item()) != 0:
                raise PeerFailure(r, int(t.item()), phase)


def guarded(fn, phase: str, group=None, device=None):
    """fn() is a local step between two transfers: its failure is announced to the peers (agree) before it is re-raised."""
    try:
        out = fn()
    except PeerFailure:
        raise
    except Exception as exc:
        agree(int(getattr(exc, "code", -1) or -1), phase, group, device)
        raise
    agree(0, phase, group, device)
    return out


# ---- whole sharded passes over torch.distributed (the harness's counterpart of mdbg_kminmer_count_first_sharded) ----------
def _device_rows(ptr: int, shape: tuple) -> torch.Tensor:
    from . import capi
    n = 1
    for s in shape:
        n *= s
    if n == 0:
        return torch.empty(shape, dtype=torch.int64, device="cuda")
    return torch.as_tensor(capi.DeviceView(ptr, shape), device="cuda")


def _exchange_shard(sh, finish, group=None):
    """rows of `sh` to their owners, mdbg_shard_reduce there, replies back; finish(device pointer of the replies) -> table."""
    rw = sh.row_words
    sent = [int(c) for c in sh.counts]
    mine, got = exchange_by_owner(_device_rows(sh.d_rows, (sh.n_rows, rw)), sent, group)
    torch.cuda.current_stream().synchronize()
    d_reply = guarded(lambda: sh.reduce(mine.data_ptr(), mine.shape[0]), "summing the rows it owns", group, "cuda")
    glob = reply_to_senders(_device_rows(d_reply, (mine.shape[0],)), got, sent, group)
    torch.cuda.current_stream().synchronize()
    table = finish(glob.data_ptr())
    sh.free()
    return table


def first_pass_sharded(ctx, reads, k: int, min_abundance: int = 0, group=None):
    """k = firstK over reads sharded across the ranks of `group`: this rank's share of the global table."""
    sh = guarded(lambda: ctx.shard_begin(reads, k, dist.get_world_size(group)), "before the exchange", group, "cuda")
    return _exchange_shard(sh, lambda d: sh.finish(d, min_abundance), group)


def allgather_records(table, group=None) -> bytes:
    """The 20-byte records of every rank's share, concatenated: the complete table of this k, which every rank loads as the
    previous table of the next (mdbg_prev_from_records).  20 bytes per key: a few hundred MB at 40 M reads."""
    rec, _ = table.to_host()
    return allgather_bytes(rec.tobytes(), group)


def allgather_bytes(data: bytes, group=None) -> bytes:
    """Every rank's byte string, concatenated in rank order, as ONE padded tensor all-gather (sizes first) -- not
    all_gather_object, which pickles hundreds of MB through the host on every rank."""
    world = dist.get_world_size(group)
    raw = torch.frombuffer(bytearray(data), dtype=torch.uint8) if len(data) else torch.empty(0, dtype=torch.uint8)
    dev = "cpu" if _staged(group) else "cuda"
    sizes = [torch.empty(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([raw.numel()], dtype=torch.int64, device=dev), group=group)
    sizes = [int(t.item()) for t in sizes]
    pad = torch.zeros(max(max(sizes), 1), dtype=torch.uint8, device=dev)
    pad[: raw.numel()] = raw.to(dev)
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return b"".join(p[:n].cpu().numpy().tobytes() for p, n in zip(parts, sizes))


def next_k_sharded(ctx, reads, unitigs, k: int, first_k: int, prev_records: bytes, group=None):
    """k > firstK (graph/CreateMdbg.cpp:391-468) over sharded reads: the ordinary refined / index pass over this rank's reads
    against the complete previous table, then mdbg_shard_from_table -> owners -> mdbg_shard_keep decide who lists the keys
    several ranks found.  `unitigs` (unitig_data.txt sequences, may be None) belong to one rank only."""
    def local_half():
        prev = ctx.prev_from_records(prev_records)
        local = ctx.kminmer_count_refined(reads, unitigs, k, prev) if k == first_k + 1 else ctx.kminmer_index(reads, unitigs, k, prev)
        return prev, local, ctx.shard_from_table(local, dist.get_world_size(group))
    prev, local, sh = guarded(local_half, "before the exchange", group, "cuda")
    out = _exchange_shard(sh, lambda d: sh.keep(d), group)
    local.free(); prev.free()
    return out
