#!/usr/bin/env python3
"""Development tool: rank 0's share of the sliced 8x8 PEPS for ONE world size, a few dozen contractions (for a kernel
trace):  python tools/peps_rank_one.py [bond] [world] [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from contractn_amd import dist as cdist  # noqa: E402

bond = int(sys.argv[1]) if len(sys.argv) > 1 else 8
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
einstr, shapes, ops = bench.peps_network(8, 8, bond)
labels, path, rep = cdist.sliced_plan(einstr, shapes, min_slices=64, max_intermediate=2 ** 28 if bond >= 16 else None)
sc = cdist.SlicedContraction(einstr, ops, labels, optimize=path, rank=0, world=world, device=0)
for _ in range(iters):
    sc.local_result()
torch.cuda.synchronize()
print("done", len(sc.my_slices))
if os.environ.get("DUMP_STEPS"):
    import numpy as np
    from contractn_amd.engine import KERNEL_NAMES
    ex = sc.bc.executor
    ex.set_timing(3)
    for _ in range(3):
        sc.local_result()
    torch.cuda.synchronize()
    ms = ex.step_ms()
    infos = sc.bc.plan.step_infos()
    tiles = ex.step_tiles()
    top = np.argsort(-ms)[:14]
    print("total event ms", float(ms.sum()))
    for i in top:
        x = infos[i]
        print(i, KERNEL_NAMES[x["kernel"]], "B", x["batch"], "M", x["m"], "N", x["n"], "K", x["k"], "modes", x["mode_a"], x["mode_b"],
              "tile", tiles[i], "blocks", x["blocks"], "ms", round(float(ms[i]), 4),
              "GB/s", round(len(sc.my_slices) * 4 * (x["batch"] * (x["m"] * x["k"] + x["k"] * x["n"] + x["m"] * x["n"])) / (ms[i] * 1e-3) / 1e9, 1))
