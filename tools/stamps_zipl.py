#!/usr/bin/env python3
"""Development tool: per-workgroup cycle stamps of one k_zip_lat launch of the headline network (R networks in flight).
Needs a `make STAMPS=1 EXP=9` library: CTN_LIB_PATH=contractn_amd/lib/libctn_hip_exp9.so python tools/stamps_zipl.py [R]"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    import bench
    from contractn_amd.einsum import BatchedContraction
    R = int(sys.argv[2])
    tn, einstr, shapes, path = bench.build_network(12, 256, 4)
    bc = BatchedContraction(einstr, shapes, np.float32, optimize=path, replicas=R)
    gen = torch.Generator(device="cuda"); gen.manual_seed(0)
    ops = [[torch.randn(s, generator=gen, device="cuda") / 16.0 for s in shapes] for _ in range(R)]
    out = torch.zeros(R, 1, device="cuda")
    launch = bc.executor.make_enqueue([t.data_ptr() for rep in ops for t in rep], [out[r].data_ptr() for r in range(R)])
    for _ in range(3):
        launch()
    bc.executor.synchronize()
    print(bc.executor.step_tiles()[:10])
else:
    R = sys.argv[1] if len(sys.argv) > 1 else "1"
    f = "/tmp/stamps_zipl.bin"
    env = dict(os.environ, CTN_DEBUG_STAMPS=f, CTN_DEBUG_STAMP_STEP="8", CTN_GRAPH="0", CTN_ZIPL="1", CTN_ZIP="0")
    subprocess.run([sys.executable, os.path.abspath(__file__), "child", R], env=env, check=True)
    a = np.fromfile(f, dtype=np.uint64).reshape(-1, 8).astype(np.int64)
    a = a[(a[:, :7] > 0).all(axis=1)]
    med = lambda x: int(np.median(x))
    names = ["requests issued", "E sums + phase 1", "T to LDS", "barrier", "phase 2", "store"]
    print("R", R, "workgroups", len(a), " ".join(f"{n}: {med(a[:, k + 1] - a[:, k])}" for k, n in enumerate(names)),
          "total", med(a[:, 6] - a[:, 0]), "span", a[:, 6].max() - a[:, 0].min(), "(cycles; MFMA ideal per phase 2048 per wave, two waves per SIMD)")
