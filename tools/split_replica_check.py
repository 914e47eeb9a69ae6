#!/usr/bin/env python3
"""Development check: the K-split forms (dot / row-dot / streaming / MFMA latency mode with slab folding) with
several replicas in flight - every replica against NumPy."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from contractn_amd import einsum as E  # noqa: E402

CASES = [("abc,abc->", [(64, 64, 64), (64, 64, 64)]), ("ab,b->a", [(100, 40000), (40000,)]),
         ("ab,ab->a", [(300, 8200), (300, 8200)]), ("ab->b", [(2048, 512)]), ("ab,ab->b", [(3000, 260), (3000, 260)]),
         ("km,kn->mn", [(19200, 64), (19200, 64)]), ("ka,kb->ab", [(100000, 8), (100000, 8)]),
         ("abk,k->ab", [(37, 44, 16), (16,)])]
bad = 0
for dtype, tol in ((np.float32, 2e-5), (np.float64, 1e-12)):
    for ein, shapes in CASES:
        rng = np.random.default_rng(5)
        R = 3
        sets = [[(rng.standard_normal(s) + 0.25).astype(dtype) * (r + 1) for s in shapes] for r in range(R)]
        path = ((0, 1),) if len(shapes) == 2 else ((0,),)
        bc = E.BatchedContraction(ein, shapes, dtype, optimize=path, replicas=R)
        t, c = bc.run_host(sets)
        for r in range(R):
            ref = np.einsum(ein, *[o.astype(np.float64) for o in sets[r]])
            terms = np.einsum(ein, *[np.abs(o).astype(np.float64) for o in sets[r]])
            got = np.asarray(t[r], dtype=np.float64) * np.exp(float(c[r]))
            err = float(np.max(np.abs(got - ref) / terms))
            if not err <= tol:
                bad += 1
                print("BAD", np.dtype(dtype).name, ein, r, err)
        bc.executor.close()
print("bad", bad)
