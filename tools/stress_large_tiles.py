"""Development tool: a few hundred random label structures through the large-tile LDS-DMA kernels (forced),
fp32 and fp64, against numpy.einsum.  `python tools/stress_large_tiles.py`"""
import os, sys
os.environ["CTN_MFMA_G"] = "2"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from contractn_amd import contract, einsum as E
from tests.test_gpu_fuzz import _large_tile_case
bad = 0
for dtype, tol in (("float32", 1e-4), ("float64", 1e-11)):
    for seed in range(300):
        rng = np.random.default_rng(20000 + seed)
        einstr, sizes = _large_tile_case(rng, dtype)
        lhs = einstr.split("->")[0].split(",")
        ops = [rng.standard_normal([sizes[c] for c in t]).astype(dtype) for t in lhs]
        ref = np.einsum(einstr, *[o.astype(np.float64) for o in ops])
        t_hat, c = contract(einstr, *ops, optimize=((0, 1),), split_format=True)
        got = t_hat.astype(np.float64) * np.exp(float(c))
        err = np.max(np.abs(got - ref)) / max(np.max(np.abs(ref)), 1e-300)
        if not (err <= tol) or got.shape != ref.shape:
            bad += 1
            print("BAD", dtype, seed, einstr, sizes, err, flush=True)
        if seed % 100 == 99:
            E.clear_caches()
            print(dtype, seed + 1, "done", flush=True)
print("bad", bad)
