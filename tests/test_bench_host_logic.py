"""bench.py's host-side logic that needs no GPU."""
from __future__ import annotations

import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_deadline_ends_a_hung_run_with_a_line():
    """MDBG_BENCH_DEADLINE_S: a run that does not finish -- a collective whose peer never arrives; RCCL with more than one rank has not met
    hardware yet -- ends with ONE JSON line that carries "value": null and names the phase, the Python stacks on stderr, exit status 3."""
    code = ("import os, sys, time; sys.path.insert(0, %r); import bench; bench._phase('waiting for a peer'); "
            "bench._arm_deadline(0, 8, os.dup(1)); time.sleep(30)" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=25, env=dict(os.environ, MDBG_BENCH_DEADLINE_S="1"))
    assert r.returncode == 3, r.stderr[-500:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["value"] is None and line["n_gpus"] == 8 and "waiting for a peer" in line["error"]
    assert "giving up" in r.stderr and "Thread" in r.stderr          # faulthandler's dump of every thread


def test_deadline_of_zero_never_fires_and_other_ranks_stay_silent_on_stdout():
    code = ("import os, sys, time; sys.path.insert(0, %r); import bench; bench._arm_deadline(3, 8, os.dup(1)); time.sleep(2.5)" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=25, env=dict(os.environ, MDBG_BENCH_DEADLINE_S="0"))
    assert r.returncode == 0 and r.stdout == ""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=25, env=dict(os.environ, MDBG_BENCH_DEADLINE_S="1"))
    assert r.returncode == 3 and r.stdout == "" and "rank 3 of 8" in r.stderr


def test_cpu_quota_is_read_from_the_cgroup():
    sys.path.insert(0, ROOT)
    import bench
    q = bench._cpu_quota()
    assert q is None or q > 0
    assert 1 <= bench._cores_used(32) <= 32
