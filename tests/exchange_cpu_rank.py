"""One rank (one PROCESS) of tests/test_exchange_on_cpu.py: the library's peer-copy exchange -- the SHIPPED csrc/multigpu.hip + peerlink.hpp +
context.hip, compiled by g++ against a stand-in for the HIP runtime over host memory (tests/host/hip_on_host) -- driven through the C ABI on a CPU.
    python tests/exchange_cpu_rank.py <lib> <rank> <n_ranks> <id file> <out .json> <k> <mode>
Every rank takes its contiguous share of one seeded minimizer-space read set; its rows [hash_lo, hash_hi, count], grouped by owner, come from
the CPU oracle (as in tests/test_distributed_gloo.py); mdbg_shard_exchange moves them to their owners, the owner-side reduction (a host
restatement of mdbg_shard_reduce, tests/host/exchange_double.cpp) answers, the replies come back; the rank checks every reply against the
counts over ALL the reads and writes what it saw.  Three exchanges over one communicator (the control block's words alternate).
Environment: MDBG_TEST_DIE_BEFORE_PASS=<rank>:<pass> (that rank's process ends without a word before the pass),
MDBG_TEST_FAIL_REDUCE=<rank>:<pass> (that rank's reduction fails in the pass)."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as orc  # noqa: E402
from tests.test_distributed_gloo import EMIT_BIT, owner_of, partial_rows  # noqa: E402

lib_path, rank, n_ranks, id_file, out, k, mode = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], int(sys.argv[6]), sys.argv[7]
L = C.CDLL(lib_path)
L.mdbg_last_error.restype = C.c_char_p
L.mdbg_last_error.argtypes = [C.c_void_p]
L.mdbg_comm_note.restype = C.c_char_p
L.mdbg_comm_note.argtypes = [C.c_void_p]
MODES = {"rccl": 0, "peer": 1, "auto": 2}


def check(ctx, rc, what):
    if rc != 0:
        raise RuntimeError(f"{what}: {rc}: {L.mdbg_last_error(ctx).decode()}")


t_start = time.time()
ctx = C.c_void_p()
check(None, L.mdbg_create(rank, C.byref(ctx)), "mdbg_create")
uid = open(id_file, "rb").read()
comm = C.c_void_p()
rc = L.mdbg_comm_create_mode(ctx, uid, rank, n_ranks, MODES[mode], C.byref(comm))
result = {"rank": rank, "create_rc": rc, "create_error": L.mdbg_last_error(ctx).decode() if rc else "", "passes": []}
if rc != 0:
    json.dump(result, open(out, "w"))
    sys.exit(4)
result["mode"] = int(L.mdbg_comm_mode(comm))
shard = C.c_void_p()
check(ctx, L.mdbg_shard_for_test(ctx, C.byref(shard)), "mdbg_shard_for_test")
die_rank, die_pass = (int(x) for x in os.environ.get("MDBG_TEST_DIE_BEFORE_PASS", "-1:-1").split(":"))
bad_rank, bad_pass = (int(x) for x in os.environ.get("MDBG_TEST_FAIL_REDUCE", "-1:-1").split(":"))
for n_pass in range(3):
    if rank == die_rank and n_pass == die_pass:
        os._exit(9)
    # one read set per pass (another seed each time: the staging buffers grow on the way), every rank its share
    rng = np.random.default_rng(70 + n_pass)
    lens = rng.integers(0, 40, 60 * n_ranks * (1 + 2 * n_pass))
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    mins = rng.integers(0, 9, int(offs[-1])).astype(np.uint32)
    per = (len(lens) + n_ranks - 1) // n_ranks
    lo_r, hi_r = min(rank * per, len(lens)), min((rank + 1) * per, len(lens))
    rows, counts = partial_rows(orc, mins[int(offs[lo_r]): int(offs[hi_r])], offs[lo_r: hi_r + 1] - offs[lo_r], k, n_ranks)
    rows = np.ascontiguousarray(rows).view(np.uint64)
    d_rows = C.c_void_p()
    check(ctx, L.mdbg_test_device_alloc(ctx, max(rows.nbytes, 8), C.byref(d_rows)), "device alloc")
    C.memmove(d_rows, rows.ctypes.data, rows.nbytes)
    cnt = (C.c_uint64 * 64)(*([int(c) for c in counts] + [0] * (64 - n_ranks)))
    if rank == bad_rank and n_pass == bad_pass:
        L.mdbg_shard_test_fail_next_reduce(shard)
    d_rep = C.c_void_p()
    t0 = time.time()
    rc = L.mdbg_shard_exchange(ctx, comm, shard, d_rows, cnt, C.byref(d_rep))
    entry = {"pass": n_pass, "rc": rc, "rows": int(len(rows)), "seconds": round(time.time() - t0, 3)}
    if rc != 0:
        entry["error"] = L.mdbg_last_error(ctx).decode()
    else:
        glob = np.ctypeslib.as_array(C.cast(d_rep, C.POINTER(C.c_uint64)), shape=(max(len(rows), 1),))[: len(rows)].copy()
        exp_rows, _ = partial_rows(orc, mins, offs, k, 1)
        exp = {(int(r[1]), int(r[0])): int(r[2]) for r in exp_rows.view(np.uint64)}
        entry["replies_right"] = bool(all(exp[(int(r[1]), int(r[0]))] == (int(g) & 0xFFFFFFFF) for r, g in zip(rows, glob)))
        entry["listed"] = int(sum(1 for g in glob if int(g) & EMIT_BIT))
        entry["keys_in_all"] = len(exp)
        entry["rows_to_owner"] = [int(c) for c in counts]
        entry["owners_right"] = bool((owner_of(rows[:, 1], n_ranks) == np.repeat(np.arange(n_ranks), counts)).all()) if len(rows) else True
    result["passes"].append(entry)
    L.mdbg_test_device_free(d_rows)
st = (C.c_uint64 * 8)()
ms = C.c_double()
L.mdbg_comm_stats(comm, st, C.byref(ms))
result["stats"] = {"exchanges": int(st[3]), "bytes_to_peers": int(st[4]), "bytes_from_peers": int(st[5]), "bytes_local": int(st[6])}
json.dump(result, open(out, "w"))
L.mdbg_comm_destroy(comm)
L.mdbg_destroy(ctx)
