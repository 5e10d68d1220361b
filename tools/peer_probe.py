"""One rank of a PROBE of the library's peer-copy transport, in a process of its own (bench.py, N > 1):
    MDBG_PROBE_ID=<hex of the 128-byte communicator id> python tools/peer_probe.py <rank> <n_ranks> <device>
Creates a library context on <device> and a communicator over peer copies (MDBG_COMM_PEER): attach, the staging buffers shared between
the processes, the self-test's pulls from every other rank's device and its agreed verdict -- everything an exchange will do, on a few
rows.  Exit status 0: the copies work between these devices.  Why a process of its own: the one thing the library cannot turn into an error
code is a GPU memory access fault on the first pull from a device this one cannot address -- it ends the process; here that process is a
child, and the job falls back to RCCL instead of dying (round-5 ADVICE: the copies had never crossed two devices when they became the default)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metamdbg_amd import capi  # noqa: E402

rank, n_ranks, device = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
if os.environ.get("MDBG_PROBE_TEST_CRASH_RANK") == str(rank):      # tests: this rank's probe dies the way a GPU memory fault ends a process
    import signal
    os.kill(os.getpid(), signal.SIGSEGV)
uid = bytes.fromhex(os.environ["MDBG_PROBE_ID"])
ctx = capi.Context(device)
comm = ctx.comm_create(uid, rank, n_ranks, "peer")
ok = comm.mode == "peer"
comm.destroy()
ctx.close()
sys.exit(0 if ok else 3)
