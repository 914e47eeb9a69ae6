#!/usr/bin/env python3
"""Development tool: per-tile cycle stamps of one GEMM launch of config 3b (needs a `make STAMPS=1` library and
CTN_DEBUG_STAMPS=<file> CTN_DEBUG_STAMP_STEP=<step>)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from contractn_amd import einsum as E  # noqa: E402
from tools.batch_streams import network  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
einstr, shapes, path = network(B)
bc = E.BatchedContraction(einstr, shapes, np.float32, optimize=path, replicas=1)
gen = torch.Generator(device="cuda"); gen.manual_seed(0)
ops = [torch.randn(s, generator=gen, device="cuda") / 4.0 for s in shapes]
out = torch.zeros((1,) + tuple(bc.plan.out_shape), device="cuda")
launch = bc.executor.make_enqueue([t.data_ptr() for t in ops], [out[0].data_ptr()])
for _ in range(3):
    launch()
bc.executor.synchronize()
infos = bc.plan.step_infos()
print([ (i, x["kernel"], x["epilogue_sum"], x["m"], x["n"], x["k"]) for i, x in enumerate(infos[:8])])
