#!/usr/bin/env python3
"""Development tool: time single pairwise steps through the C ABI (HIP events), one line per case.

    python tools/gemm_lab.py "mk,kn->mn:256,1024,256:128" "km,kn->mn:256,256,1024:128" ...

case = einsum:extent of each distinct label in order of first appearance:replicas[:dtype]
"""
import sys
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from contractn_amd.einsum import BatchedContraction  # noqa: E402


def run(case, iters=20):
    parts = case.split(":")
    ein, dims, R = parts[0], [int(x) for x in parts[1].split(",")], int(parts[2])
    dtype = np.dtype(parts[3]) if len(parts) > 3 else np.dtype(np.float32)
    labels = []
    for ch in ein.replace(",", "").replace("->", ""):
        if ch not in labels:
            labels.append(ch)
    size = dict(zip(labels, dims))
    terms = ein.split("->")[0].split(",")
    shapes = [tuple(size[c] for c in t) for t in terms]
    bc = BatchedContraction(ein, shapes, dtype, optimize=((0, 1),) if len(terms) == 2 else ((0,),), replicas=R)
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    ops = [[torch.randn(s, device="cuda", dtype=tdt) for s in shapes] for _ in range(R)]
    out = torch.zeros((R,) + bc.plan.out_shape, device="cuda", dtype=tdt)
    launch = bc.executor.make_enqueue([t.data_ptr() for r in ops for t in r], [out[r].data_ptr() for r in range(R)])
    for _ in range(3):
        launch()
    bc.executor.synchronize()
    bc.executor.set_timing(iters)
    for _ in range(iters):
        launch()
    ms = float(bc.executor.step_ms()[0])
    info = bc.plan.step_info(0)
    tf = info["flops"] * R / (ms * 1e-3) / 1e12
    # spot check against torch on replica 0
    ref = torch.einsum(ein, *[o.double() for o in ops[0]])
    scale = ref.abs().mean()
    err = float(((out[0].double() * scale) - ref).abs().max() / ref.abs().max())
    print(f"{case:48s} kernel={info['kernel']} modes=({info['mode_a']},{info['mode_b']}) blocks/rep={info['blocks']:5d} "
          f"{ms*1e3:9.1f} us {tf:7.1f} TF  err={err:.1e}")


if __name__ == "__main__":
    for c in sys.argv[1:]:
        run(c)
