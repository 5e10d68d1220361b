"""The control block of the library's peer-copy exchange (metamdbg_amd/csrc/peerlink.hpp) on a CPU: ranks as processes and as threads."""
from __future__ import annotations

import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("peerlink") / "test_peerlink")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", os.path.join(ROOT, "tests", "host", "test_peerlink.cpp"), "-o", out, "-lpthread", "-lrt"], check=True)
    return out


@pytest.mark.parametrize("ranks,exchanges", [(2, 400), (8, 150)])
def test_phases_statuses_deadlines(exe, ranks, exchanges):
    """Hundreds of three-phase exchanges with every rank reading every rank's words after every wait; a local failure seen by all at its
    phase and the next exchange in step again; a rank that never arrives named after the deadline; a rank that left noticed; no name
    left under /dev/shm."""
    r = subprocess.run([exe, str(ranks), str(exchanges)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.startswith("ok:"), r.stdout + r.stderr


def test_under_thread_sanitizer(tmp_path):
    """The same with -fsanitize=thread (threads of one process share the block through separate mappings; the forked cases run too)."""
    out = str(tmp_path / "test_peerlink_tsan")
    b = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", os.path.join(ROOT, "tests", "host", "test_peerlink.cpp"), "-o", out, "-lpthread", "-lrt"],
                       capture_output=True, text=True)
    if b.returncode != 0:
        pytest.skip("no thread sanitizer runtime here")
    r = subprocess.run([out, "3", "60"], capture_output=True, text=True, timeout=300, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
    if "FATAL: ThreadSanitizer" in r.stderr and "unexpected memory mapping" in r.stderr:
        pytest.skip("the thread sanitizer cannot run in this container (address space layout)")
    assert r.returncode == 0 and "WARNING: ThreadSanitizer" not in r.stderr, r.stdout + r.stderr[-3000:]
