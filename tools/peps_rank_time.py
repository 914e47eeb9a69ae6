#!/usr/bin/env python3
"""Development tool: the compute part of the PEPS strong-scaling curve on ONE GPU - rank 0's share of the slices
for world = 1, 2, 4, 8 (no process group: `SlicedContraction.local_result`, host read-back included), i.e. what one of
N GPUs would spend per contraction before the join.

    python tools/peps_rank_time.py [bond] [slices]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from contractn_amd import dist as cdist  # noqa: E402

bond = int(sys.argv[1]) if len(sys.argv) > 1 else 8
slices = int(sys.argv[2]) if len(sys.argv) > 2 else 64
einstr, shapes, ops = bench.peps_network(8, 8, bond)
labels, path, rep = cdist.sliced_plan(einstr, shapes, min_slices=slices, max_intermediate=2 ** 28 if bond >= 16 else None)
base = None
for world in (1, 2, 4, 8):
    sc = cdist.SlicedContraction(einstr, ops, labels, optimize=path, rank=0, world=world, device=0)
    for _ in range(4):
        sc.local_result()
    torch.cuda.synchronize()
    iters = 20 if bond < 16 else 2
    t0 = time.perf_counter()
    for _ in range(iters):
        sc.local_result()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    base = base or ms
    print(json.dumps({"bond": bond, "slices_total": rep["slices"], "world": world, "slices_on_rank0": len(sc.my_slices),
                      "ms_per_contraction_rank0": round(ms, 3), "speedup_vs_world1_compute_only": round(base / ms, 2)}), flush=True)
    del sc
    torch.cuda.empty_cache()
