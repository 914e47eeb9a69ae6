#!/usr/bin/env python3
"""Development tool: the PEPS boundary absorption (64 x 32768 x 64 per slice, 64 slices) in every operand layout -
which orientation / memory order of the big tensor does the register-staged kernel stream fastest?"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

import config_bench as cb  # noqa: E402
from contractn_amd.engine import KERNEL_NAMES  # noqa: E402

R, BIG, S = 64, 32768, 64
cases = [
    ("km,kn->mn", [(S, S), (S, BIG)]),     # small^T x big[k][n]     -> [m][n]   (rows of 128 KB)
    ("mk,kn->mn", [(S, S), (S, BIG)]),
    ("km,nk->mn", [(S, S), (BIG, S)]),     # big[n][k] (k-contiguous) -> [m][n]
    ("nk,km->nm", [(BIG, S), (S, S)]),     # big[n][k] as the ROW operand -> [n][m]: every tile a contiguous block
    ("nk,mk->nm", [(BIG, S), (S, S)]),
    ("kn,km->nm", [(S, BIG), (S, S)]),     # big[k][n] as the row operand -> [n][m]
]
for einstr, shapes in cases:
    bc, ops, out, wall, ms, by = cb.run_device(einstr, shapes, ((0, 1),), replicas=R, iters=10, scale=8.0)
    i = bc.plan.step_infos()[0]
    nbytes = R * 4 * (S * S + 2 * S * BIG)
    print(json.dumps({"einsum": einstr, "kernel": KERNEL_NAMES[i["kernel"]], "swapped": i["swapped"], "m": i["m"], "n": i["n"],
                      "modes": [i["mode_a"], i["mode_b"]], "tile": bc.executor.step_tiles()[0], "us": round(float(ms[0]) * 1e3, 1),
                      "GB/s": round(nbytes / (float(ms[0]) * 1e-3) / 1e9, 1)}), flush=True)
    del ops, out, bc
    torch.cuda.empty_cache()
