#!/usr/bin/env python3
"""Development tool: CP through a copy node (BASELINE config 4 (i) r = n = 1024 and (iv) r = 4096), Khatri-Rao product
fused into the GEMM's A operand (default above 2^28 elements) against materialised (CTN_FUSE=0): ms per contraction.
    python tools/cp_fuse_probe.py [r ...]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from contractn_amd.einsum import BatchedContraction  # noqa: E402

n = 1024
for r in [int(x) for x in sys.argv[1:]] or [1024, 4096]:
    shapes = [(r, n)] * 3
    bc = BatchedContraction("ac,ad,ae->cde", shapes, np.float32, optimize="auto", replicas=1)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    ops = [torch.randn(s, generator=gen, device="cuda") / 32.0 for s in shapes]
    res = torch.empty(tuple(bc.plan.out_shape), device="cuda")
    torch.cuda.synchronize()
    launch = bc.executor.make_enqueue([t.data_ptr() for t in ops], [res.data_ptr()])
    for _ in range(2):
        launch()
    bc.executor.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        launch()
    bc.executor.synchronize()
    sec = (time.perf_counter() - t0) / 3
    bc.executor.set_timing(2)
    for _ in range(2):
        launch()
    bc.executor.synchronize()
    ms = bc.executor.step_ms()
    infos, tiles = bc.plan.step_infos(), bc.executor.step_tiles()
    c = float(bc.fetch_log_scale()[0])
    e = (5, 700, 1023)
    ref = float((ops[0][:, e[0]].double() * ops[1][:, e[1]].double() * ops[2][:, e[2]].double()).sum())
    got = float(res[e]) * float(np.exp(c))
    print(json.dumps({"r": r, "CTN_FUSE": os.environ.get("CTN_FUSE"), "ms": round(sec * 1e3, 3),
                      "tflops": round(bc.plan.flops / sec / 1e12, 2), "frac": round(bc.plan.flops / sec / 1e12 / 157.3, 4),
                      "steps": [(i["kernel"], i["mode_a"], i["mode_b"], tl, round(float(m), 3)) for i, tl, m in zip(infos, tiles, ms)],
                      "spot_rel_err": abs(got - ref) / abs(ref)}), flush=True)
    bc.executor.close()
    del ops, res
    torch.cuda.empty_cache()
