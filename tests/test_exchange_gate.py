"""metamdbg_amd.distributed.ExchangeGate (bench.py's N > 1 pipeline): scans of the batches in flight and the exchange of one of them exclude
each other; nothing deadlocks, with the exchanges also taking turns in global step order as bench.py makes them.  Threads and sleeps only."""
from __future__ import annotations

import random
import threading
import time

import pytest

from metamdbg_amd.distributed import ExchangeGate


def _pipeline(n_slots: int, n_steps: int, enabled: bool, seed: int):
    """bench.py's step(): scan -> local half -> (turn) exchange -> finish, step i on slot i % n_slots; returns the log of intervals."""
    gate = ExchangeGate(enabled)
    turn = threading.Condition()
    next_exchange = [0]
    log = []                       # (kind, start, end)
    lock = threading.Lock()
    errors = []

    def run(slot):
        rng = random.Random(seed * 100 + slot)
        try:
            for i in range(slot, n_steps, n_slots):
                with gate.scan():
                    t0 = time.perf_counter()
                    time.sleep(rng.uniform(0.0005, 0.003))
                    t1 = time.perf_counter()
                with lock:
                    log.append(("scan", t0, t1))
                time.sleep(rng.uniform(0.0, 0.002))                     # purge + the local half
                with turn:
                    assert turn.wait_for(lambda: next_exchange[0] >= i, timeout=20), "turn never came"
                try:
                    with gate.exchange():
                        t0 = time.perf_counter()
                        time.sleep(rng.uniform(0.0002, 0.001))
                        t1 = time.perf_counter()
                    with lock:
                        log.append(("exchange", t0, t1))
                finally:
                    with turn:
                        next_exchange[0] = i + 1
                        turn.notify_all()
                time.sleep(rng.uniform(0.0, 0.0005))                    # finish
        except BaseException as exc:
            errors.append(exc)
            with turn:
                next_exchange[0] = 1 << 60
                turn.notify_all()

    threads = [threading.Thread(target=run, args=(s,)) for s in range(n_slots)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=60)
    assert not any(t.is_alive() for t in threads), "deadlock"
    assert not errors, errors
    return gate, log


def _overlaps(log):
    scans = [(a, b) for k, a, b in log if k == "scan"]
    return sum(1 for k, a, b in log if k == "exchange" for (c, d) in scans if a < d and c < b)


@pytest.mark.parametrize("n_slots", [1, 2, 3, 4])
def test_no_scan_runs_beside_an_exchange(n_slots):
    gate, log = _pipeline(n_slots, 120, True, seed=n_slots)
    assert sum(1 for k, *_ in log if k == "scan") == sum(1 for k, *_ in log if k == "exchange") == 120
    assert _overlaps(log) == 0
    if n_slots > 1:
        assert gate.waited_ms > 0.0          # some exchange had to wait for a scan in flight


def test_a_disabled_gate_lets_them_overlap():
    gate, log = _pipeline(3, 150, False, seed=9)
    assert _overlaps(log) > 0 and gate.waited_ms == 0.0


def test_a_scan_that_fails_releases_the_gate():
    gate = ExchangeGate(True)
    with pytest.raises(RuntimeError):
        with gate.scan():
            raise RuntimeError("scan failed")
    done = []
    t = threading.Thread(target=lambda: (gate.exchange().__enter__(), done.append(1)))
    t.start()
    t.join(timeout=5)
    assert done == [1]


def test_an_exchange_that_fails_lets_the_scans_go_on():
    gate = ExchangeGate(True)
    with pytest.raises(RuntimeError):
        with gate.exchange():
            raise RuntimeError("peer failed")
    done = []
    t = threading.Thread(target=lambda: (gate.scan().__enter__(), done.append(1)))
    t.start()
    t.join(timeout=5)
    assert done == [1]


def test_two_exchanges_at_once_do_not_block_each_other():
    """(bench.py never has two -- they take turns -- but a dead slot releases the turn for good, and the survivors must still get through)"""
    gate = ExchangeGate(True)
    inside = threading.Barrier(2, timeout=5)

    def ex():
        with gate.exchange():
            inside.wait()
    ts = [threading.Thread(target=ex) for _ in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=10)
    assert not any(t.is_alive() for t in ts)
