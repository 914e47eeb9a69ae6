#!/usr/bin/env python3
"""Development tool: the compute part of the PEPS strong-scaling curve on ONE GPU, for the STAGED sliced plan that
`bench.py --config peps` runs (`dist.StagedSlicedContraction`).  For world = 1, 2, 4, 8 EVERY rank's share is built and
timed on the one GPU, one rank after another (no process group: `run()` with the cross-rank all_gather left out, local
combine and the copy of the 16-byte result to the host included) - i.e. what each of N GPUs would spend per contraction
before the join.  One JSON line per (world, rank), then one summary line per world with

    predicted_speedup = best_single_gpu_ms / (max_g ms_g + join_us / 1000)

where `join_us` is what ONE all_gather of 16 bytes over 8 ranks is ASSUMED to cost (RCCL small-message latency over
xGMI; 50 us, on the safe side - it cannot be measured on a one-GPU box) and `best_single_gpu_ms` the faster of the
staged plan with all slices on one GPU and - where it fits - the unsliced network on the library's own path.

    python tools/peps_rank_time.py [bond] [worlds, e.g. 1,2,4,8] [log2 of the bond-16 plan's cap: 32 | 28]  >  profiles/r04_peps_D<bond>_rank_time.jsonl
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from contractn_amd import dist as cdist  # noqa: E402
from contractn_amd.einsum import BatchedContraction  # noqa: E402

JOIN_US_ASSUMED = 50.0

bond = int(sys.argv[1]) if len(sys.argv) > 1 else 8
worlds = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4, 8]
big = bond >= 16
einstr, shapes, ops = bench.peps_network(8, 8, bond)
cap = 2 ** int(sys.argv[3]) if len(sys.argv) > 3 else 2 ** 32          # bond 16: the plan's largest tensor (elements)
labels, path, rep = cdist.staged_plan(einstr, shapes, min_slices=64, max_intermediate=cap if big else None)
dev = torch.device("cuda", 0)
warm, iters = (1, 1) if big else (4, 20)

unsliced_ms = None
if rep["unsliced_largest_intermediate"] <= 2 ** 28:
    bc = BatchedContraction(einstr, shapes, np.float32, optimize="auto", replicas=1, device=0)
    d_ops = [torch.as_tensor(o, device=dev) for o in ops]
    res = torch.zeros(1, device=dev)
    torch.cuda.synchronize()
    launch = bc.executor.make_enqueue([t.data_ptr() for t in d_ops], [res.data_ptr()])
    for _ in range(4):
        launch()
    bc.executor.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        launch()
    bc.executor.synchronize()
    unsliced_ms = (time.perf_counter() - t0) / 20 * 1e3
    bc.executor.close()
    del bc, d_ops

best_single = None
value0 = None
for world in worlds:
    per_rank = []
    for rank in range(world):
        sc = cdist.StagedSlicedContraction(einstr, ops, labels, optimize=path, rank=rank, world=world, device=0)
        sc.world = 1                      # the local part only: no collective (there is no process group here)
        for _ in range(warm):
            t, c = sc.run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            t, c = sc.run()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / iters * 1e3
        evals = [[len(dep), n] for dep, n, _all in sc.evaluations()]
        launches = sum(st[3] * st[0].plan.n_steps for st in sc.stage_list())
        per_rank.append(ms)
        print(json.dumps({"bond": bond, "world": world, "rank": rank, "rank_grid": list(sc.rank_grid),
                          "slices_on_rank": len(sc.my_slices), "slices_total": sc.n_total,
                          "ms_per_contraction": round(ms, 4), "steps_enqueued_per_contraction": launches,
                          "stage_evaluations_[n_labels,evals]": evals, "part": [float(t), float(c)]}), flush=True)
        if world == 1:
            value0 = (float(t), float(c))
        del sc
        torch.cuda.empty_cache()
    if world == 1:
        best_single = min(x for x in (per_rank[0], unsliced_ms) if x is not None)
    if best_single is not None:
        worst = max(per_rank)
        print(json.dumps({"bond": bond, "world": world, "summary": True, "max_rank_ms": round(worst, 4),
                          "min_rank_ms": round(min(per_rank), 4), "mean_rank_ms": round(float(np.mean(per_rank)), 4),
                          "staged_single_gpu_ms": None if world != 1 else round(per_rank[0], 4),
                          "unsliced_single_gpu_ms": None if unsliced_ms is None else round(unsliced_ms, 4),
                          "best_single_gpu_ms": round(best_single, 4), "join_us_assumed": JOIN_US_ASSUMED,
                          "predicted_speedup": round(best_single / (worst + JOIN_US_ASSUMED / 1e3), 3),
                          "predicted_speedup_compute_only": round(best_single / worst, 3)}), flush=True)
