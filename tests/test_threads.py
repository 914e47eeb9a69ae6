"""Re-entrancy of ``contract`` on the host side (CPU only: the native executor is replaced by a probe).

The reference's ``contract`` may be called from several threads at once (einsum.py:190-310 holds no mutable
state but the path cache).  A native executor owns one workspace, one pointer table and one graph capture, and
ctypes releases the GIL inside it, so the host layer must never let two threads drive the same executor
concurrently: executors are cached per calling thread and every use holds the executor's lock."""
import threading
import time

import numpy as np

from contractn_amd import einsum as E
from contractn_amd import engine


class ProbeExecutor:
    """Stands in for engine.Executor: records how many threads are inside it at once."""

    created = []

    def __init__(self, plan, replicas=1, device=0, stream=None):
        self.plan, self.replicas = plan, replicas
        self.lock = threading.RLock()
        self.inside = 0
        self.max_inside = 0
        self.users = set()
        self.closed = False
        ProbeExecutor.created.append(self)

    def run_host(self, operand_sets):
        assert not self.closed, "executor used after close()"
        held = self.lock.acquire(blocking=False)      # the caller must already hold the lock (RLock: re-entrant)
        assert held, "run_host called without holding the executor's lock"
        try:
            self.inside += 1
            self.max_inside = max(self.max_inside, self.inside)
            self.users.add(threading.get_ident())
            time.sleep(0.002)                          # the native call releases the GIL for about this long
            self.inside -= 1
        finally:
            self.lock.release()
        outs = np.zeros((self.replicas,) + self.plan.out_shape, dtype=self.plan.np_dtype)
        return outs, np.zeros(self.replicas), np.ones((self.replicas, self.plan.n_steps))

    def close(self):
        with self.lock:
            self.closed = True


def test_two_threads_never_share_an_executor_concurrently(monkeypatch):
    E.clear_caches()
    ProbeExecutor.created = []
    monkeypatch.setattr(engine, "Executor", ProbeExecutor)
    a, b = np.ones((4, 5)), np.ones((5, 6))
    clist = E._contract_path("ab,bc->ac", ((4, 5), (5, 6)), optimize="auto", memory_limit=None, use_blas=True)
    errors = []
    start = threading.Barrier(4)

    def worker():
        try:
            start.wait()
            for _ in range(25):
                t, c = E._core_contract([a, b], clist, "numpy")
                assert t.shape == (4, 6)
        except Exception as exc:  # noqa: BLE001 - reported below
            errors.append(exc)

    threads = [threading.Thread(target=worker) for _ in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert len(ProbeExecutor.created) == 4                      # one executor per calling thread ...
    assert all(len(ex.users) == 1 for ex in ProbeExecutor.created)
    assert all(ex.max_inside == 1 for ex in ProbeExecutor.created)  # ... and never two threads inside one
    E.clear_caches()
    assert all(ex.closed for ex in ProbeExecutor.created)


def test_lru_eviction_closes_executors_and_is_bounded(monkeypatch):
    E.clear_caches()
    ProbeExecutor.created = []
    monkeypatch.setattr(engine, "Executor", ProbeExecutor)
    monkeypatch.setattr(E, "MAX_CACHED_EXECUTORS", 3)
    for n in range(2, 8):
        a, b = np.ones((n, 3)), np.ones((3, 2))
        clist = E._contract_path("ab,bc->ac", ((n, 3), (3, 2)), optimize="auto", memory_limit=None, use_blas=True)
        E._core_contract([a, b], clist, "numpy")
    assert len(ProbeExecutor.created) == 6
    assert sum(not ex.closed for ex in ProbeExecutor.created) == 3
    E.clear_caches()
    assert all(ex.closed for ex in ProbeExecutor.created)


def test_an_executor_evicted_between_lookup_and_lock_is_looked_up_again(monkeypatch):
    """Eviction is global, the cache per thread: with more live (thread, plan) pairs than slots, another thread's insertion
    can close the executor a thread has just looked up.  Many threads on a two-slot cache, each with its own plan: no run
    ever reaches a closed executor (the probe asserts that), every result arrives."""
    E.clear_caches()
    ProbeExecutor.created = []

    class ClosableProbe(ProbeExecutor):
        def is_open(self):
            return not self.closed

    monkeypatch.setattr(engine, "Executor", ClosableProbe)
    monkeypatch.setattr(E, "MAX_CACHED_EXECUTORS", 2)
    errors = []
    start = threading.Barrier(6)

    def worker(n):
        try:
            a, b = np.ones((n, 3)), np.ones((3, 2))
            clist = E._contract_path("ab,bc->ac", ((n, 3), (3, 2)), optimize="auto", memory_limit=None, use_blas=True)
            start.wait()
            for _ in range(40):
                t, _c = E._core_contract([a, b], clist, "numpy")
                assert t.shape == (n, 2)
        except Exception as exc:  # noqa: BLE001 - reported below
            errors.append(exc)

    threads = [threading.Thread(target=worker, args=(n,)) for n in range(2, 8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert len(ProbeExecutor.created) > 6                       # evictions did happen
    E.clear_caches()


def test_torch_operands_that_require_grad_are_refused():
    """The reference's torch backend is differentiable; this engine is not - it must say so instead of
    returning a result without a graph."""
    import pytest
    import torch

    x = torch.ones(3, 4, requires_grad=True)
    y = torch.ones(4, 5)
    with pytest.raises(NotImplementedError, match="autograd"):
        E.contract("ab,bc->ac", x, y)
