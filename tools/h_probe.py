"""Per-launch time of the batched-MPS site step (config 3b) with and without the one-tile-per-CU form (CTN_H)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch, time
    from contractn_amd import TN
    from contractn_amd.einsum import BatchedContraction
    from contractn_amd.paths import ssa_to_linear
    from tests import networks as nets
    B, n_sites, bond, phys = int(sys.argv[2]), 100, 256, 4
    class Shape:
        def __init__(self, shape): self.shape, self.ndim = tuple(shape), len(shape)
    tn = TN(); hub = tn.add_copy_node(n_sites + 1)
    cores = [Shape((phys, bond) if i in (0, n_sites - 1) else (phys, bond, bond)) for i in range(n_sites)]
    nodes = nets.add_mps(tn, cores)
    for i, node in enumerate(nodes):
        inp = tn.add_input_node((B, phys), var_shape_axes=(0,)); tn.connect_nodes(inp, node, 1, 0); tn.connect_nodes(hub, inp, i, 0)
    shapes = [c.shape for c in cores] + [(B, phys)] * n_sites
    path = ssa_to_linear(nets.batched_mps_path(n_sites), 2 * n_sites)
    bc = BatchedContraction(tn.einsum_str, shapes, np.float32, optimize=path, replicas=1, device=0)
    gen = torch.Generator(device="cuda"); gen.manual_seed(11)
    ops = [torch.randn(sh, generator=gen, device="cuda") / 4.0 for sh in shapes]
    res = torch.zeros((1,) + tuple(bc.plan.out_shape), device="cuda")
    launch = bc.executor.make_enqueue([t.data_ptr() for t in ops], [res[0].data_ptr()])
    for _ in range(4): launch()
    bc.executor.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): launch()
    bc.executor.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    bc.executor.set_timing(3)
    for _ in range(3): launch()
    bc.executor.synchronize()
    st = bc.executor.step_ms().astype(float)
    infos = bc.plan.step_infos(); tiles = bc.executor.step_tiles()
    dom = [i for i, x in enumerate(infos) if x["epilogue_sum"] > 0]
    print(json.dumps({"B": B, "CTN_H": os.environ.get("CTN_H"), "ms_per_pass": round(ms, 4), "tflops": round(bc.plan.flops / ms / 1e9, 2),
                      "site_launch_us": round(float(np.mean(st[dom])) * 1e3, 2), "tile": tiles[dom[1]], "checksum": float(res.abs().sum())}))
else:
    for B in (4096, 2048, 8192):
        for h in ("0", None):
            env = dict(os.environ)
            if h is not None: env["CTN_H"] = h
            else: env.pop("CTN_H", None)
            subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(B)], env=env)
