#!/usr/bin/env python3
"""Development tool: per-tile cycle stamps of one site launch of config 3b on the one-tile-per-CU form (k_mfma_f32_h).
Needs a `make STAMPS=1 EXP=9` library: CTN_LIB_PATH=contractn_amd/lib/libctn_hip_exp9.so python tools/stamps_h.py"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from contractn_amd import einsum as E
    from tools.batch_streams import network
    einstr, shapes, path = network(4096)
    bc = E.BatchedContraction(einstr, shapes, np.float32, optimize=path, replicas=1)
    gen = torch.Generator(device="cuda"); gen.manual_seed(0)
    ops = [torch.randn(s, generator=gen, device="cuda") / 4.0 for s in shapes]
    out = torch.zeros((1,) + tuple(bc.plan.out_shape), device="cuda")
    launch = bc.executor.make_enqueue([t.data_ptr() for t in ops], [out[0].data_ptr()])
    for _ in range(3):
        launch()
    bc.executor.synchronize()
    print(bc.executor.step_tiles()[int(os.environ["CTN_DEBUG_STAMP_STEP"])])
else:
    f = "/tmp/stamps_h.bin"
    env = dict(os.environ, CTN_DEBUG_STAMPS=f, CTN_DEBUG_STAMP_STEP="22", CTN_GRAPH="0")
    subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=True)
    a = np.fromfile(f, dtype=np.uint64).reshape(-1, 8).astype(np.int64)
    a = a[(a[:, :4] > 0).all(axis=1)]
    med = lambda x: int(np.median(x))
    print("tiles", len(a), "prologue", med(a[:, 1] - a[:, 0]), "main", med(a[:, 2] - a[:, 1]), "exchange", med(a[:, 4] - a[:, 2]),
          "epilogue math+stores", med(a[:, 5] - a[:, 4]), "block_sum+tail", med(a[:, 3] - a[:, 5]), "total", med(a[:, 3] - a[:, 0]),
          "span", a[:, 3].max() - a[:, 0].min())
