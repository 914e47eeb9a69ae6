#!/usr/bin/env python3
"""Development tool: per-workgroup cycle stamps of the k_sweep_f32 launch of the batched MPS (BASELINE config 3b).
Needs a `make STAMPS=1 EXP=n` library: CTN_LIB_PATH=contractn_amd/lib/libctn_hip_expn.so python tools/stamps_sweep.py [B] [sites]
Timing only (any library): python tools/stamps_sweep.py child B sites [bond] [phys]   (CTN_SWEEP=0: the per-site launches)"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from contractn_amd import TN
    from contractn_amd.einsum import BatchedContraction
    from contractn_amd.paths import ssa_to_linear
    from tests import networks as nets
    B, n_sites = int(sys.argv[2]), int(sys.argv[3])
    bond = int(sys.argv[4]) if len(sys.argv) > 4 else 256
    phys = int(sys.argv[5]) if len(sys.argv) > 5 else 4
    tn, inputs = nets.batched_mps(TN, n_sites, bond, phys, 16, dtype=np.float32, seed=4)
    shapes = [p.shape for p in tn.params] + [(B, phys)] * n_sites
    path = ssa_to_linear(nets.batched_mps_path(n_sites), 2 * n_sites)
    bc = BatchedContraction(tn.einsum_str, shapes, np.float32, optimize=path, replicas=1)
    gen = torch.Generator(device="cuda"); gen.manual_seed(0)
    ops = [torch.randn(s, generator=gen, device="cuda") / (4.0 if len(s) == 2 and s[0] == B else bond ** 0.5) for s in shapes]
    out = torch.zeros((1,) + tuple(bc.plan.out_shape), device="cuda")
    if os.environ.get("SWEEP_ALIAS"):     # every interior core the same buffer: the stream comes out of L2 for sure
        for i in range(2, n_sites - 1):
            ops[i] = ops[1]
    launch = bc.executor.make_enqueue([t.data_ptr() for t in ops], [out[0].data_ptr()])
    for _ in range(3):
        launch()
    bc.executor.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(10):
        launch()
    bc.executor.synchronize()
    print("ms per pass", (time.perf_counter() - t0) * 100.0)
    tiles = bc.executor.step_tiles()
    print([s for s, t in enumerate(tiles) if t[0] == 16 and t[1] >= 128])
else:
    B = sys.argv[1] if len(sys.argv) > 1 else "4096"
    sites = sys.argv[2] if len(sys.argv) > 2 else "100"
    f = "/tmp/stamps_sweep.bin"
    env = dict(os.environ, CTN_DEBUG_STAMPS=f, CTN_GRAPH="0")
    subprocess.run([sys.executable, os.path.abspath(__file__), "child", B, sites], env=env, check=True)
    a = np.fromfile(f, dtype=np.uint64).reshape(-1, 8).astype(np.int64)
    a = a[a[:, 0] > 0]
    med = lambda x: int(np.median(x))
    n = int(sites) - 2
    print("workgroups", len(a), "sites", n, "wave 0: total", med(a[:, 1] - a[:, 0]), "k-loops", med(a[:, 2]), "epilogues", med(a[:, 3]),
          "per site", med(a[:, 2]) // n, "+", med(a[:, 3]) // n, "| wave 7: k-loops", med(a[:, 6]), "epilogues", med(a[:, 7]),
          "| ideal k-loop per site", 512 * 32 * 2, "| span", (a[:, 1].max() - a[:, 0].min()))
