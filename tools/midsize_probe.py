#!/usr/bin/env python3
"""Development tool: one network in flight (R = 1), 100-site MPS overlap at several bond dimensions -
wall time per pass of the enqueue loop (no per-step events) next to the sum of per-step device times,
to see whether the host's launch rate or the device's per-kernel latency bounds mid-size networks."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from contractn_amd.einsum import BatchedContraction  # noqa: E402


def main():
    bonds = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [8, 16, 32, 64, 128, 256]
    for dtype in (np.float32, np.float64):
        for D in bonds:
            tn, einstr, shapes, path = bench.build_network(100, D, 4)
            bc = BatchedContraction(einstr, shapes, dtype, optimize=path, replicas=1)
            tdt = torch.float32 if dtype == np.float32 else torch.float64
            ops = [torch.randn(s, device="cuda", dtype=tdt) / 4 for s in shapes]
            out = torch.zeros((1,) + tuple(bc.plan.out_shape), device="cuda", dtype=tdt)
            launch = bc.executor.make_enqueue([t.data_ptr() for t in ops], [out[0].data_ptr()])
            for _ in range(3):
                launch()
            bc.executor.synchronize()
            n = 30
            t0 = time.perf_counter()
            for _ in range(n):
                launch()
            t_host = (time.perf_counter() - t0) / n
            bc.executor.synchronize()
            wall = (time.perf_counter() - t0) / n
            bc.executor.set_timing(5)
            for _ in range(5):
                launch()
            bc.executor.synchronize()
            dev = float(bc.executor.step_ms().sum())
            ns = bc.plan.n_steps
            kinds = {}
            for i in bc.plan.step_infos():
                kinds[i["kernel"]] = kinds.get(i["kernel"], 0) + 1
            print(f"{np.dtype(dtype).name} D={D:4d} steps={ns} wall {wall*1e3:7.3f} ms ({wall*1e6/ns:6.2f} us/step)  "
                  f"host enqueue {t_host*1e3:7.3f} ms  device sum {dev:7.3f} ms  kernels {kinds}", flush=True)
            bc.executor.close()


if __name__ == "__main__":
    main()
