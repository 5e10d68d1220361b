"""Two, four and eight real ranks of bench.py's sharded step on ONE GPU (every rank on device 0; torch.distributed on gloo -- RCCL refuses two
ranks per device -- the rows moved by the library's peer copies or, staged through the host, by torch.distributed): the whole multi-rank orchestration (shard begin / reduce / finish, the two
all-to-alls, two batches in flight taking turns on the communicator) must give, summed over the ranks, exactly the
table of one rank over the same reads."""
from __future__ import annotations

import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_with_a_free_port(make_cmd, **kw):
    """torch.distributed.run on a port found free a moment ago; the moment can be too long (another process takes the port: "EADDRINUSE" once in
    some hundred runs of the suite) -- then once more on another port."""
    for attempt in range(3):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        out = subprocess.run(make_cmd(port), **kw)
        if "EADDRINUSE" not in (out.stderr or "") or attempt == 2:
            return out
    return out


def _bench(extra_env: dict, args: list[str], nproc: int | None, expect_failure: bool = False) -> dict:
    """One bench.py run; returns its FULL result (--detail) after checking that what it printed is the one compact line of it."""
    import tempfile
    env = dict(os.environ, **extra_env)
    detail = tempfile.NamedTemporaryFile(suffix=".json", delete=False).name
    args = list(args) + ["--detail", detail]
    def make_cmd(port):
        c = [sys.executable]
        if nproc:
            c += ["-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1", "--master-port", str(port)]
        return c + [os.path.join(ROOT, "bench.py")] + args
    out = _run_with_a_free_port(make_cmd, env=env, capture_output=True, text=True, timeout=280, cwd=ROOT)
    cmd = make_cmd(0)
    if (out.returncode != 0) != expect_failure:
        # torchrun's summary of who was killed fills the tail: what the ranks themselves said is further up -- kept whole for the record
        os.makedirs(os.path.join(ROOT, "gpurun_out", "test_failures"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "test_failures", f"bench_{os.getpid()}_{len(os.listdir(os.path.join(ROOT, 'gpurun_out', 'test_failures')))}.stderr"), "w") as f:
            f.write(" ".join(cmd) + "\n" + repr(extra_env) + "\n" + out.stderr)
        said = [ln for ln in out.stderr.splitlines() if any(w in ln for w in ("mdbg", "Mdbg", "[bench]", "Error", "error", "rank ")) and "SIGTERM" not in ln]
        raise AssertionError(f"bench.py ended with status {out.returncode}:\n" + "\n".join(said[-40:]))
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 4096, out.stdout[-500:]
    line = json.loads(lines[0])
    with open(detail) as f:
        full = json.load(f)
    os.unlink(detail)
    assert line["value"] == pytest.approx(full["value"], rel=1e-5) and line["n_gpus"] == full["n_gpus"]
    assert (line["config"].get("exchange") or {}).get("transport") == (full["config"].get("exchange") or {}).get("transport")
    return full


@pytest.mark.parametrize("nproc,in_flight", [(2, 1), (2, 2), (2, 3), (4, 2), (8, 2)])
def test_ranks_equal_one(nproc, in_flight):
    """2, 4 and 8 real processes (round-3 VERDICT: more than two had only ever been emulated inside one process): the 8 x 8 count
    matrix, the owner map at 8 and the turn-taking of the batches in flight on every rank, against one rank over the same reads."""
    n = 40_000
    common = ["--steps", "3", "--warmup", "1", "--cpu-sample", "0", "--legs", "none", "--in-flight", str(in_flight)]
    one = _bench({}, ["--gpus", "1", "--reads", str(nproc * n)] + common, None)
    # (the default exchange of an N > 1 job is the library's, which takes peer copies here: RCCL refuses two ranks on one device; the
    # first case keeps torch.distributed's all-to-all, staged through the host by gloo, covered)
    env = {"MDBG_BENCH_SHARE_GPU": "1", "MDBG_BENCH_BACKEND": "gloo"}
    if (nproc, in_flight) == (2, 1):
        env["MDBG_BENCH_EXCHANGE"] = "torch"
    many = _bench(env, ["--gpus", str(nproc), "--reads", str(n)] + common, nproc)
    assert many["config"]["exchange"]["transport"] == ("torch" if "MDBG_BENCH_EXCHANGE" in env else "peer")
    assert many["n_gpus"] == nproc and many["scaling"] == "weak"
    assert many["config"]["kminmer_records"] == one["config"]["kminmer_records"] > 0
    assert many["config"]["solid"] == one["config"]["solid"] > 0
    # the N > 1 line checks itself: rank 0 repeats the pass alone over all the reads and compares with the summed shares
    par = many["parity"]
    assert par["table_equal"] and par["reads"] == nproc * n and par["records_equal"] and par["abundance_checksum_equal"]
    assert par["single_gpu"]["records"] == one["config"]["kminmer_records"]
    assert par["single_gpu_gbps"] > 0 and par["single_gpu_pass_seconds"] > 0
    ex = many["config"]["exchange"]
    assert ex["ranks"] == nproc and ex["wire_bytes_per_step"] > 0 and ex["exchanges_timed"] == nproc * 3 and ex["exchange_ms_per_step"] > 0


@pytest.mark.parametrize("in_flight", [2, 3])
def test_ranks_equal_one_with_the_exchange_gate(in_flight):
    """MDBG_BENCH_EXCHANGE_GATE=1 (the default of an N > 1 job over RCCL, forced here on the gloo path): a batch about to exchange waits for
    the scans in flight on its rank and holds new ones back.  Same tables, no rank left waiting (the run ends), the line says the gate was on."""
    n, nproc = 40_000, 4
    common = ["--steps", "4", "--warmup", "1", "--cpu-sample", "0", "--legs", "none", "--in-flight", str(in_flight)]
    one = _bench({}, ["--gpus", "1", "--reads", str(nproc * n)] + common, None)
    many = _bench({"MDBG_BENCH_SHARE_GPU": "1", "MDBG_BENCH_BACKEND": "gloo", "MDBG_BENCH_EXCHANGE_GATE": "1", "MDBG_BENCH_EXCHANGE": "torch"},
                  ["--gpus", str(nproc), "--reads", str(n)] + common, nproc)
    assert many["config"]["exchange"]["gate"] is True and many["config"]["exchange"]["exchanges_timed"] == nproc * 4
    assert many["config"]["kminmer_records"] == one["config"]["kminmer_records"] > 0
    assert many["parity"]["table_equal"] and many["parity"]["reads"] == nproc * n


@pytest.mark.parametrize("nproc,in_flight", [(2, 2), (4, 3), (8, 2)])
def test_ranks_equal_one_by_peer_copies(nproc, in_flight):
    """MDBG_COMM_MODE=peer: the two all-to-alls as peer copies INSIDE THE LIBRARY (csrc/multigpu.hip, mdbg_comm_create_mode + mdbg_shard_exchange):
    every rank's staging buffers shared with the other processes by hipIpcGetMemHandle / hipIpcOpenMemHandle, owners pull their slices
    device to device on one stream per peer, hand-shakes through the shared control block (csrc/peerlink.hpp).  Real processes, real IPC
    handles (all on GPU 0 here; between GPUs the same copies cross xGMI); the union of the shares must be the table of one rank."""
    n = 40_000
    common = ["--steps", "4", "--warmup", "1", "--cpu-sample", "0", "--legs", "none", "--in-flight", str(in_flight)]
    one = _bench({}, ["--gpus", "1", "--reads", str(nproc * n)] + common, None)
    many = _bench({"MDBG_BENCH_SHARE_GPU": "1", "MDBG_BENCH_BACKEND": "gloo", "MDBG_COMM_MODE": "peer"},
                  ["--gpus", str(nproc), "--reads", str(n)] + common, nproc)
    ex = many["config"]["exchange"]
    assert ex["transport"] == "peer" and ex["path"].startswith("library: peer copies") and ex["gate"] is False and ex["exchanges_timed"] == nproc * 4
    assert ex["wire_bytes_per_step"] > 0 and ex["comm_note"] is None
    assert many["config"]["kminmer_records"] == one["config"]["kminmer_records"] > 0 and many["config"]["solid"] == one["config"]["solid"]
    par = many["parity"]
    assert par["table_equal"] and par["reads"] == nproc * n and par["abundance_checksum_equal"] and par["vector_sum_equal"]


def test_peer_copies_with_a_corrupted_reply_fail_the_run():
    n = 40_000
    common = ["--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--legs", "none", "--in-flight", "2"]
    many = _bench({"MDBG_BENCH_SHARE_GPU": "1", "MDBG_BENCH_BACKEND": "gloo", "MDBG_COMM_MODE": "peer", "MDBG_BENCH_CORRUPT_REPLY": "1"},
                  ["--gpus", "2", "--reads", str(n)] + common, 2, expect_failure=True)
    assert many["config"]["exchange"]["transport"] == "peer"
    assert not many["parity"]["table_equal"] and many["parity"]["minimizers_equal"]


def test_auto_falls_back_to_the_next_transport_on_every_rank(tmp_path):
    """MDBG_COMM_MODE=auto with the control block out of reach (MDBG_PEER_TEST_NO_SHM: rank 1 cannot attach; the others wait for it until the
    set-up deadline): every rank gives up the peer copies TOGETHER and tries RCCL, which refuses ranks that share a device -- so the job, all
    ranks agreeing once more, moves its rows with torch.distributed, says so, and its tables still add up."""
    n = 40_000
    common = ["--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--legs", "none", "--in-flight", "1"]
    many = _bench({"MDBG_BENCH_SHARE_GPU": "1", "MDBG_BENCH_BACKEND": "gloo", "MDBG_COMM_MODE": "auto", "MDBG_PEER_TEST_NO_SHM": "1", "MDBG_PEER_SETUP_TIMEOUT_S": "3"},
                  ["--gpus", "2", "--reads", str(n)] + common, 2)
    ex = many["config"]["exchange"]
    assert ex["transport"] == "torch" and "library exchange unavailable" in ex["comm_note"]
    assert many["parity"]["table_equal"]


def test_a_probe_that_dies_sends_every_rank_to_the_next_transport():
    """bench.py tries the peer copies in CHILD processes before the ranks themselves create communicators over them (tools/peer_probe.py): what
    the library cannot turn into an error code -- a GPU memory access fault on the first pull from a device that cannot be addressed -- ends a
    child.  Here rank 1's probe is killed by a signal: every rank then takes RCCL (which refuses ranks that share a device, so the job, agreeing
    once more, moves its rows with torch.distributed), says why, and its tables still add up."""
    n = 40_000
    common = ["--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--legs", "none", "--in-flight", "2"]
    many = _bench({"MDBG_BENCH_SHARE_GPU": "1", "MDBG_BENCH_BACKEND": "gloo", "MDBG_PROBE_TEST_CRASH_RANK": "1", "MDBG_PEER_SETUP_TIMEOUT_S": "5"}, ["--gpus", "3", "--reads", str(n)] + common, 3)
    ex = many["config"]["exchange"]
    assert ex["transport"] == "torch" and "probe failed" in ex["comm_note"] and "library exchange unavailable" in ex["comm_note"], ex
    assert many["parity"]["table_equal"]


def test_strong_scaling_over_one_read_set():
    """--total-reads: ONE read set split over the ranks (north_star's "40 M reads sharded 8 ways at 1 / 2 / 4 / 8 GPUs" is a curve over a
    fixed set): 4 ranks x 30 000 reads give the table of 1 rank x 120 000, and the line carries the single-GPU throughput of rank 0's
    verification pass -- the N = 1 point of that curve."""
    total = 120_000
    common = ["--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--legs", "none", "--total-reads", str(total)]
    one = _bench({}, ["--gpus", "1"] + common, None)
    four = _bench({"MDBG_BENCH_SHARE_GPU": "1", "MDBG_BENCH_BACKEND": "gloo"}, ["--gpus", "4"] + common, 4)
    assert one["scaling"] == four["scaling"] == "strong"
    assert one["config"]["reads_per_gpu"] == total and four["config"]["reads_per_gpu"] == total // 4
    assert four["config"]["kminmer_records"] == one["config"]["kminmer_records"] > 0
    assert four["parity"]["table_equal"] and four["parity"]["reads"] == total and four["parity"]["single_gpu_gbps"] > 0


@pytest.mark.parametrize("nproc", [2, 8])
def test_ranks_with_a_corrupted_reply_fail_the_run(nproc):
    """MDBG_BENCH_CORRUPT_REPLY=1: one global count of the verification step is off by one on the last rank -- the line still comes
    out, its parity block says the tables do not add up, and the run exits non-zero."""
    n = 40_000
    common = ["--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--legs", "none", "--in-flight", "2"]
    many = _bench({"MDBG_BENCH_SHARE_GPU": "1", "MDBG_BENCH_BACKEND": "gloo", "MDBG_BENCH_CORRUPT_REPLY": "1"},
                  ["--gpus", str(nproc), "--reads", str(n)] + common, nproc, expect_failure=True)
    par = many["parity"]
    assert not par["table_equal"] and par["minimizers_equal"]
    assert not (par["abundance_checksum_equal"] and par["sum_abundance_equal"] and par["solid_equal"] and par["records_equal"])


@pytest.mark.parametrize("exchange", ["library", "torch"])
def test_a_rank_that_fails_in_the_reduction_ends_the_job_of_eight(exchange):
    """MDBG_BENCH_FAIL_RANK=5: rank 5 of 8 fails summing the rows it owns (the second phase of an exchange) in the verification step.
    Every rank hears of it in the agreement that follows -- the status words of the library's peer copies (csrc/multigpu.hip peer_phase),
    or metamdbg_amd/distributed.py agree / guarded on the torch.distributed path -- and the job ends non-zero within seconds: nobody
    waits for replies that will never come."""
    import time
    n = 20_000
    env = dict(os.environ, MDBG_BENCH_SHARE_GPU="1", MDBG_BENCH_BACKEND="gloo", MDBG_BENCH_FAIL_RANK="5", MDBG_BENCH_DEADLINE_S="200", MDBG_BENCH_EXCHANGE=exchange,
               MDBG_COMM_MODE="peer")
    def make_cmd(port):
        return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1", "--master-port", str(port),
                os.path.join(ROOT, "bench.py"), "--gpus", "8", "--reads", str(n), "--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--legs", "none"]
    t0 = time.time()
    out = _run_with_a_free_port(make_cmd, env=env, capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert out.returncode != 0 and time.time() - t0 < 200
    own = "test failure in the reduction" if exchange == "library" else "test failure on rank 5"
    assert own in out.stderr and "rank 5 failed summing the rows it owns" in out.stderr, out.stderr[-3000:]
    assert "deadline" not in out.stderr and "did not arrive" not in out.stderr       # ended by the protocol, not by a watchdog


def _library_exchange_processes(tmp_path, n_ranks: int, mode: str, share: bool, n_total: int = 4000):
    import numpy as np
    from metamdbg_amd import capi, formats, synth
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "comm_rank.py"), str(r), str(n_ranks), str(tmp_path / "id"),
                               str(tmp_path / f"rec{r}.npy"), str(n_total), mode, "1" if share else "0"], cwd=ROOT) for r in range(n_ranks)]
    for p in procs:
        assert p.wait(timeout=240) == 0
    rec = np.concatenate([np.load(tmp_path / f"rec{r}.npy") for r in range(n_ranks)])
    modes = {open(tmp_path / f"rec{r}.npy.mode").read() for r in range(n_ranks)}
    ctx = capi.Context(0)
    spec = synth.hifi_spec(n_total, seed=23, read_len=6000, coverage=25.0)
    corr = ctx.purge_palindromes(ctx.scan(ctx.reads_synthetic(spec), K=15, density=0.005, hpc=True), 4, 100)
    one, _ = ctx.kminmer_count_first(corr, 4, 0).to_host()
    assert np.array_equal(formats.sorted_abundance_records(rec), formats.sorted_abundance_records(one))
    return modes


@pytest.mark.parametrize("n_ranks", [2, 3, 8])
def test_library_peer_exchange_processes_sharing_one_gpu(tmp_path, n_ranks):
    """The library's peer-copy exchange through the C ABI alone -- no torch.distributed anywhere: n processes, the communicator id in a
    file, mdbg_comm_create_mode(MDBG_COMM_PEER) + mdbg_kminmer_count_first_sharded three times over.  Staging buffers cross the process
    boundary by hipIpcGetMemHandle / hipIpcOpenMemHandle (all ranks on GPU 0 here); the union of the shares is the single-GPU table."""
    assert _library_exchange_processes(tmp_path, n_ranks, "peer", share=True) == {"peer"}


def test_a_rank_that_dies_ends_its_peers_at_the_deadline(tmp_path):
    """No protocol can be agreed on with a process that is gone.  Rank 1 of 3 ends without a word before its second pass; its peers, already
    inside that pass's exchange, wait for it at the first phase -- and return MDBG_EPEER naming it when MDBG_PEER_TIMEOUT_S has passed, instead
    of hanging (an RCCL receive would).  The communicator is broken from then on."""
    import time
    env = dict(os.environ, MDBG_TEST_DIE_BEFORE_PASS="1:1", MDBG_PEER_TIMEOUT_S="4")
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "comm_rank.py"), str(r), "3", str(tmp_path / "id"), str(tmp_path / f"rec{r}.npy"), "3000", "peer", "1"],
                              cwd=ROOT, env=env, stderr=subprocess.PIPE, text=True) for r in range(3)]
    outs = [p.communicate(timeout=200) for p in procs]
    assert [p.returncode for p in procs] == [3, 9, 3] and time.time() - t0 < 120
    for r in (0, 2):
        assert "pass 1 failed" in outs[r][1] and "rank 1 did not arrive" in outs[r][1] and "within 4 s" in outs[r][1], outs[r][1][-800:]


@pytest.mark.parametrize("mode", ["rccl", "peer", "auto"])
def test_library_exchange_two_gpus(tmp_path, mode):
    """Two ranks, two GPUs, the exchange over xGMI inside the library -- RCCL send / receive groups, or peer copies between the
    two devices -- (mdbg_comm_create_mode + mdbg_kminmer_count_first_sharded): the union of the two tables equals the single-GPU
    table of all reads.  Skipped on a box with one GPU (one rank: tests/test_gpu_parity.py::test_library_exchange_one_rank; several
    ranks on one device, peer copies only: above)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    assert _library_exchange_processes(tmp_path, 2, mode, share=False) == {"rccl" if mode == "rccl" else "peer"}


def test_two_ranks_multik_equal_one(tmp_path):
    """BASELINE.json configs[2] on more than one rank: the multi-k loop k = 4 .. 8 (benchmark mode: reads only, previous table =
    the gathered records of k - 1) with the reads sharded over two real ranks (gloo, both on GPU 0) against one rank doing all the
    reads: the gathered records must be the same table at every k."""
    import socket as _s
    import numpy as np
    from metamdbg_amd import capi, formats, synth
    n_total, last_k = 6000, 8
    r = _run_with_a_free_port(lambda port: [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                                            "--master-port", str(port), os.path.join(ROOT, "tests", "rank_multik.py"), str(tmp_path), str(n_total), str(last_k)],
                              capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    ctx = capi.Context(0)
    spec = synth.hifi_spec(n_total, seed=77, read_len=6000, coverage=25.0)
    corr = ctx.purge_palindromes(ctx.scan(ctx.reads_synthetic(spec), K=15, density=0.005, hpc=True), 4, 100)
    t = ctx.kminmer_count_first(corr, 4, 0)
    for k in range(4, last_k + 1):
        if k > 4:
            t = ctx.kminmer_count_refined(corr, None, k, t) if k == 5 else ctx.kminmer_index(corr, None, k, t)
        rec, _ = t.to_host()
        got = np.load(tmp_path / f"k{k}.npy")
        assert len(rec) > 100 and np.array_equal(formats.sorted_abundance_records(got), formats.sorted_abundance_records(rec)), k
