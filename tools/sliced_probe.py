#!/usr/bin/env python3
"""Development tool: 8 x 8 PEPS (D = 8) sliced over bonds chosen together with the path
(`dist.choose_slices_with_path`), all slices of one GPU as replicas of one plan; next to the unsliced run."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from contractn_amd import TN, dist  # noqa: E402
from contractn_amd import einsum as E  # noqa: E402
from tests import networks as nets  # noqa: E402


def main():
    rows = cols = 8
    bond = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    n_slices = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    rng = np.random.default_rng(6)
    tn = nets.peps_closed(TN, rows, cols, 2, dtype=np.float32, seed=6)
    shapes = [tuple(bond if (d == 2 and p.ndim > 1 and ax > 0) else d for ax, d in enumerate(p.shape)) for p in tn.params]
    ops = [(rng.standard_normal(s) / bond ** 0.5).astype(np.float32) for s in shapes]
    t0 = time.perf_counter()
    labels, path, rep = dist.choose_slices_with_path(tn.einsum_str, shapes, min_slices=n_slices)
    rep["search_s"] = round(time.perf_counter() - t0, 1)
    print(json.dumps({"labels": len(labels), **rep}), flush=True)
    sc = dist.SlicedContraction(tn.einsum_str, ops, labels, optimize=path, rank=0, world=1)
    sc.run()
    t0 = time.perf_counter()
    iters = 5
    for _ in range(iters):
        t_s, c_s = sc.run()
    wall = (time.perf_counter() - t0) / iters
    print(json.dumps({"config": f"PEPS {rows}x{cols} D={bond} sliced with path re-optimisation", "slices": sc.n_total,
                      "replicas_per_launch": sc.R, "ms_per_contraction": round(wall * 1e3, 3),
                      "tflops": round(sc.bc.plan.flops * sc.n_total / wall / 1e12, 2), "result": [float(t_s), float(c_s)]}),
          flush=True)
    bc = E.BatchedContraction(tn.einsum_str, shapes, np.float32, optimize="auto", replicas=1)
    outs, logs = bc.run_host([ops])
    print(json.dumps({"unsliced_result": [float(outs[0]), float(logs[0])]}), flush=True)


if __name__ == "__main__":
    main()
