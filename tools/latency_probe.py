#!/usr/bin/env python3
"""Development tool: wall-clock latency of TN.contract-style calls for the latency-bound configs
(BASELINE configs 1 and 2) through the public API, next to the CPU oracle."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from contractn_amd import TN  # noqa: E402
from oracle import cpu_ref  # noqa: E402


def timeit(fn, n=20):
    fn()
    fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e3, np.min(ts) * 1e3


def main():
    # config 1: README copy-tensor example
    tn = TN()
    hub = tn.add_copy_node(101)
    for i in range(100):
        tn.connect_nodes(hub, tn.add_dense_node(np.array([1, 0.99])), i, 0)
    fun = tn.make_contract_fun()
    params = tn.params
    print("cfg1 copy101   gpu  median/min ms:", timeit(lambda: fun(params, ())), fun(params, ()))
    ein = tn.einsum_str
    print("cfg1 copy101   cpu  median/min ms:", timeit(lambda: cpu_ref.contract(ein, *params)))
    # config 2: 1000 x (3x3) chain, split format
    tn = TN()
    prev = tn.add_dense_node(np.ones(3))
    for _ in range(1000):
        mat = tn.add_dense_node(np.ones((3, 3)))
        tn.connect_nodes(prev, mat, -1, 0)
        prev = mat
    fun = tn.make_contract_fun(split_format=True)
    params = tn.params
    med, mn = timeit(lambda: fun(params, ()))
    print(f"cfg2 chain1000 gpu  median/min ms: {med:.3f} {mn:.3f}  ({mn:.3f} us/step)", fun(params, ()))
    ein = tn.einsum_str
    med, mn = timeit(lambda: cpu_ref.contract(ein, *params, split_format=True), n=5)
    print(f"cfg2 chain1000 cpu  median/min ms: {med:.3f} {mn:.3f}  ({mn:.3f} us/step)")


if __name__ == "__main__":
    main()


def device_time():
    """Device-side time of the chain walk (HIP events) for config 2 at R = 1 and R = 256."""
    from contractn_amd.einsum import BatchedContraction

    tn = TN()
    prev = tn.add_dense_node(np.ones(3))
    for _ in range(1000):
        mat = tn.add_dense_node(np.ones((3, 3)))
        tn.connect_nodes(prev, mat, -1, 0)
        prev = mat
    shapes = [p.shape for p in tn.params]
    for R in (1, 256):
        bc = BatchedContraction(tn.einsum_str, shapes, np.float64, replicas=R)
        sets = [list(tn.params)] * R
        bc.run_host(sets)
        bc.executor.set_timing(5)
        t0 = time.perf_counter()
        for _ in range(5):
            bc.run_host(sets)
        wall = (time.perf_counter() - t0) / 5
        ms = bc.executor.step_ms()
        print(f"   wall per call {wall*1e3:.3f} ms")
        print(f"cfg2 device time R={R}: {ms.sum()*1e3:.1f} us per walk = {ms.sum():.4f} us/step; "
              f"{R / (ms.sum() * 1e-3):.0f} contractions/s")


if __name__ == "__main__" and "--device" in sys.argv:
    device_time()
