"""world_size-2 gloo test (CPU) of the exchange steps of the sharded first pass (metamdbg_amd/distributed.py):
local count rows -> all-to-all by owner -> owner sums -> all-to-all of the global counts back to the senders;
every sent row must come back with its count over the union of the reads.  The per-rank rows are produced
here by the CPU oracle (test infrastructure); on the GPU box they come from mdbg_shard_begin / _reduce
(tests/test_gpu_parity.py::test_sharded_first_pass_on_one_gpu)."""
from __future__ import annotations

import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402


def owner_of(hi: np.ndarray, n_ranks: int) -> np.ndarray:
    """Same owner function as csrc/kminmer.hip (owner_of): top 32 bits of hash_hi scaled to [0, n_ranks)."""
    return (((hi >> np.uint64(32)) * np.uint64(n_ranks)) >> np.uint64(32)).astype(np.int64)


def partial_rows(orc, mins, offs, k, n_ranks):
    """(rows int64[n, 3] = [hash_lo, hash_hi, count] grouped by owner, counts per owner) of one shard."""
    rw = 3
    keys = {}
    for r in range(len(offs) - 1):
        m = mins[int(offs[r]): int(offs[r + 1])]
        for i in range(len(m) - k + 1):
            rev, vec, hi, lo = orc.kminmer_normalize_hash(m[i:i + k])
            e = keys.setdefault((hi, lo), [0, vec])
            e[0] += 1
    rows = np.zeros((len(keys), rw), dtype=np.uint64)
    for j, ((hi, lo), (c, vec)) in enumerate(keys.items()):
        rows[j, 0], rows[j, 1], rows[j, 2] = lo, hi, c
    own = owner_of(rows[:, 1], n_ranks) if len(rows) else np.zeros(0, np.int64)
    order = np.argsort(own, kind="stable")
    counts = [int((own == r).sum()) for r in range(n_ranks)]
    return rows[order].view(np.int64), counts


EMIT_BIT = 1 << 63


def owner_reply(rows: np.ndarray) -> np.ndarray:
    """What mdbg_shard_reduce answers: for every received row the sum of the counts of its key, with bit 63 set on
    exactly one row per key (the sender of that row lists the key)."""
    u = rows.view(np.uint64)
    tot, first = {}, {}
    for i, row in enumerate(u):
        key = (int(row[1]), int(row[0]))
        tot[key] = tot.get(key, 0) + int(row[2])
        first.setdefault(key, i)
    out = [tot[(int(r[1]), int(r[0]))] | (EMIT_BIT if first[(int(r[1]), int(r[0]))] == i else 0) for i, r in enumerate(u)]
    return np.array(out, dtype=np.uint64).view(np.int64)


def _worker(rank, world, port, k, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from metamdbg_amd import distributed as D
    from oracle import pyoracle as orc
    rng = np.random.default_rng(7)                      # same data on every rank, each takes its shard
    lens = rng.integers(0, 40, 120)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    mins = rng.integers(0, 9, int(offs[-1])).astype(np.uint32)
    lo_r, hi_r = rank * 60, (rank + 1) * 60
    soffs = offs[lo_r: hi_r + 1] - offs[lo_r]
    smins = mins[int(offs[lo_r]): int(offs[hi_r])]
    rows, counts = partial_rows(orc, smins, soffs, k, world)
    mine, got = D.exchange_by_owner(torch.from_numpy(rows.copy()), counts)
    mine_np = mine.numpy()
    if len(mine_np):
        assert (owner_of(mine_np.view(np.uint64)[:, 1], world) == rank).all()     # only keys this rank owns arrive
    reply = owner_reply(mine_np)
    glob = D.reply_to_senders(torch.from_numpy(reply), got, counts).numpy()
    # expected: counts over ALL reads, for every row this rank sent, in the order sent
    exp_rows, _ = partial_rows(orc, mins, offs, k, 1)
    exp = {(int(r[1]), int(r[0])): int(r[2]) for r in exp_rows.view(np.uint64)}
    sent = rows.view(np.uint64)
    glob = glob.view(np.uint64)
    ok = len(glob) == len(sent) and all(exp[(int(r[1]), int(r[0]))] == int(g) & 0xFFFFFFFF for r, g in zip(sent, glob))
    n_listed = int(sum(1 for g in glob if int(g) & EMIT_BIT))
    # the owners' key sets partition the global key set
    owned = {(int(r[1]), int(r[0])) for r in mine_np.view(np.uint64)}
    ok = ok and owned == {key for key in exp if owner_of(np.array([key[0]], dtype=np.uint64), world)[0] == rank}
    q.put((rank, bool(ok), len(owned), n_listed))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("k", [4, 5])
def test_exchange_reduce_reply_world2(k):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, k, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res
    assert res[0][2] > 0 and res[1][2] > 0
    assert res[0][3] + res[1][3] == res[0][2] + res[1][2]        # every key is listed by exactly one rank


# ---- sharded k > firstK: every rank runs the ordinary pass over its reads against the whole previous table; the ranks only
# agree on who lists a key several of them found (mdbg_shard_from_table -> mdbg_shard_reduce -> mdbg_shard_keep on the GPU box;
# the oracle and numpy stand in for the device here) ----
def _worker_next_k(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from metamdbg_amd import distributed as D, formats
    from oracle import pyoracle as orc
    rng = np.random.default_rng(19)
    lens = rng.integers(0, 45, 160)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    genome = rng.permutation(300).astype(np.uint32)
    mins = np.concatenate([genome[a:a + n] for a, n in zip(rng.integers(0, 250, len(lens)), lens)]).astype(np.uint32)
    per = len(lens) // world
    lo_r, hi_r = rank * per, (rank + 1) * per
    soffs = offs[lo_r: hi_r + 1] - offs[lo_r]
    smins = mins[int(offs[lo_r]): int(offs[hi_r])]
    records = orc.table_abundance_records(orc.kminmer_count_first(mins, offs, 4, 0)).tobytes()     # the complete k = 4 table
    ok = True
    sizes = []
    for k in (5, 6, 7):
        prev = orc.PrevAbundance(records); prev.overlay_unitigs([], k - 1)
        fn = orc.kminmer_count_refined if k == 5 else orc.kminmer_index
        local = orc.table_abundance_records(fn(smins, soffs, k, prev))
        rows = np.zeros((len(local), 3), dtype=np.uint64)
        rows[:, 0], rows[:, 1], rows[:, 2] = local["lo"], local["hi"], local["abundance"]
        own = owner_of(rows[:, 1], world) if len(rows) else np.zeros(0, np.int64)
        order = np.argsort(own, kind="stable")
        counts = [int((own == r).sum()) for r in range(world)]
        sent = rows[order]
        mine, got = D.exchange_by_owner(torch.from_numpy(sent.view(np.int64).copy()), counts)
        reply = owner_reply(mine.numpy())                      # bit 63 on the first row of every key
        glob = D.reply_to_senders(torch.from_numpy(reply), got, counts).numpy().view(np.uint64)
        kept = sent[(glob >> np.uint64(63)) == 1]
        mine_rec = np.zeros(len(kept), dtype=formats.ABUNDANCE_DTYPE)
        mine_rec["lo"], mine_rec["hi"], mine_rec["abundance"] = kept[:, 0], kept[:, 1], kept[:, 2].astype(np.uint32)
        records = D.allgather_bytes(mine_rec.tobytes())        # the complete table of this k on every rank
        # expected: the single-rank pass over all the reads, previous table = the single-rank table of k - 1
        if k == 5:
            exp_prev = orc.table_abundance_records(orc.kminmer_count_first(mins, offs, 4, 0)).tobytes()
        p2 = orc.PrevAbundance(exp_prev); p2.overlay_unitigs([], k - 1)
        exp = orc.table_abundance_records(fn(mins, offs, k, p2))
        exp_prev = exp.tobytes()
        ok = ok and np.array_equal(formats.sorted_abundance_records(records), formats.sorted_abundance_records(exp))
        sizes.append((len(local), len(kept), len(exp)))
    q.put((rank, bool(ok), sizes))
    dist.barrier()
    dist.destroy_process_group()


def test_next_k_dedup_by_owner_world2():
    """k = 5 (refined), 6 and 7 (index) with the reads sharded over two ranks equal the single-rank tables; keys found by both
    ranks are listed by exactly one."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_next_k, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res
    for i in range(3):
        local0, kept0, exp = res[0][2][i]
        local1, kept1, _ = res[1][2][i]
        assert exp > 20 and kept0 + kept1 == exp and local0 + local1 > exp, res     # overlapping keys, each listed once


# ---- failure behaviour of the exchange protocol (include/mdbg_hip.h "Failure behaviour of the collective calls"; the library runs
# the same agreement over RCCL inside mdbg_shard_exchange, metamdbg_amd/distributed.py over torch.distributed) ----
def _worker_failures(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    from metamdbg_amd import distributed as D

    class LocalError(RuntimeError):
        code = -3

    seen = []
    for phase, failing in (("before the exchange", 1), ("summing the rows it owns", 0), ("before the exchange", None)):
        def local_step():
            if failing == rank:
                raise LocalError("local failure")
            return 7
        try:
            D.guarded(local_step, phase)
            seen.append("ok")
        except D.PeerFailure as pf:
            seen.append(("peer", pf.rank, pf.code, pf.phase))
        except LocalError:
            seen.append("local")
        # whoever failed, both ranks are in step again: a full exchange works afterwards
        rows = torch.arange(6, dtype=torch.int64).reshape(2, 3) + 100 * rank
        mine, got = D.exchange_by_owner(rows, [1, 1])
        back = D.reply_to_senders(mine[:, 0].contiguous(), got, [1, 1])
        seen.append(back.tolist())
    # the padded byte all-gather with unequal and empty parts
    joined = D.allgather_bytes(b"" if rank == 0 else b"xyz" * 5)
    q.put((rank, seen, joined))
    dist.barrier()
    dist.destroy_process_group()


def test_local_failures_reach_the_peers_world2():
    """A rank that fails in a local step between two transfers announces it (agree / guarded): it re-raises its own error, the
    other rank gets PeerFailure naming it, nobody waits in the next all-to-all, and the following exchange works."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_failures, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict((r[0], r[1:]) for r in (q.get(timeout=120) for _ in procs))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    s0, s1 = res[0][0], res[1][0]
    assert s0[0] == ("peer", 1, -3, "before the exchange") and s1[0] == "local"
    assert s0[2] == "local" and s1[2] == ("peer", 0, -3, "summing the rows it owns")
    assert s0[4] == "ok" and s1[4] == "ok"
    for i in (1, 3, 5):                 # the exchanges in between: each rank gets back the first word of the rows it sent
        assert s0[i] == [0, 3] and s1[i] == [100, 103]
    assert res[0][1] == res[1][1] == b"xyz" * 5
