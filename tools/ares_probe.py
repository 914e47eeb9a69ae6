#!/usr/bin/env python3
"""Development tool: time the wide absorption step (256 x N x 256, N = 2^LOGN) alone, HIP events through the C ABI.

    [CTN_ARES=0] [CTN_LIB_PATH=...] python tools/ares_probe.py [LOGN] [R]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from contractn_amd.einsum import BatchedContraction  # noqa: E402

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 22
R = int(sys.argv[2]) if len(sys.argv) > 2 else 1
N = 1 << logn
for ein, shapes in (("kn,km,n->m", [(256, N), (256, 256), (N,)]), ("nk,mk,n->m", [(N, 256), (256, 256), (N,)]),
                    ("kn,mk,n->m", [(256, N), (256, 256), (N,)]), ("nk,km,n->m", [(N, 256), (256, 256), (N,)]),
                    # the wide operand on the left: the result is laid out [n][m], a tile is one contiguous piece
                    ("km,kn,n->m", [(256, 256), (256, N), (N,)]), ("mk,nk,n->m", [(256, 256), (N, 256), (N,)]),
                    ("mk,kn,n->m", [(256, 256), (256, N), (N,)]), ("km,nk,n->m", [(256, 256), (N, 256), (N,)])):
    bc = BatchedContraction(ein, shapes, np.float32, optimize=((0, 1), (0, 1)), replicas=R)
    ops = [[torch.randn(s, device="cuda") for s in shapes] for _ in range(R)]
    out = torch.zeros((R,) + bc.plan.out_shape, device="cuda")
    launch = bc.executor.make_enqueue([t.data_ptr() for r in ops for t in r], [out[r].data_ptr() for r in range(R)])
    for _ in range(2):
        launch()
    bc.executor.synchronize()
    iters = 5
    bc.executor.set_timing(iters)
    for _ in range(iters):
        launch()
    ms = float(bc.executor.step_ms()[0])
    info = bc.plan.step_info(0)
    tile = bc.executor.step_tiles()[0]
    ref = torch.einsum(ein, *[o.double() for o in ops[0]])
    got = out[0].double()
    err = float((got / got.abs().max() - ref / ref.abs().max()).abs().max())       # (the log scale aside)
    print(f"{ein:14s} N=2^{logn} R={R} modes=({info['mode_a']},{info['mode_b']}) tile={tile} {ms:8.3f} ms "
          f"{info['flops'] * R / ms / 1e9:7.1f} TF  err={err:.1e} M={info['m']} N={info['n']}", flush=True)
    del bc, ops, out
