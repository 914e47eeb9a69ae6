#!/usr/bin/env python3
"""Secondary measurements: BASELINE configs 1, 2, 3b, 4, 5 on one MI355X (one JSON line each).

    python tools/config_bench.py [cfg1 cfg2 cfg3b cfg4 cfg5]

Device time from HIP events around every step (ctn_exec_set_timing); tensors resident in HBM.
Sizes follow SURVEY.md 8(d) (configs 4 and 5 in their feasible renditions)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from contractn_amd import TN  # noqa: E402
from contractn_amd import einsum as E  # noqa: E402
from contractn_amd.engine import KERNEL_NAMES  # noqa: E402
from contractn_amd.paths import ssa_to_linear  # noqa: E402
from tests import networks as nets  # noqa: E402


class Shape:
    """Stand-in tensor: only .shape/.ndim matter for building a TN whose data lives on the GPU."""

    def __init__(self, shape):
        self.shape, self.ndim = tuple(shape), len(shape)


def run_device(einstr, shapes, path, replicas=1, iters=5, seed=0, scale=1.0, dtype=np.float32):
    bc = E.BatchedContraction(einstr, shapes, dtype, optimize=path, replicas=replicas)
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    ops = [[torch.randn(s, generator=gen, device="cuda", dtype=tdt) / scale for s in shapes] for _ in range(replicas)]
    out = torch.zeros((replicas,) + tuple(bc.plan.out_shape), device="cuda", dtype=tdt)
    launch = bc.executor.make_enqueue([t.data_ptr() for r in ops for t in r], [out[r].data_ptr() for r in range(replicas)])
    launch()
    launch()
    bc.executor.synchronize()
    t0 = time.perf_counter()           # wall time of plain enqueues (hipGraph replay, no per-step events)
    for _ in range(iters):
        launch()
    bc.executor.synchronize()
    wall = (time.perf_counter() - t0) / iters
    bc.executor.set_timing(iters)      # then the per-kernel breakdown from event-bracketed enqueues
    for _ in range(iters):
        launch()
    bc.executor.synchronize()
    ms = bc.executor.step_ms()
    infos = bc.plan.step_infos()
    by = {}
    for i, m in zip(infos, ms):
        d = by.setdefault(KERNEL_NAMES[i["kernel"]], {"ms": 0.0, "flops": 0.0, "steps": 0})
        d["ms"] += float(m); d["flops"] += i["flops"] * replicas; d["steps"] += 1
    return bc, ops, out, wall, ms, by


def summarize(name, bc, wall, by, replicas, extra=None):
    flops = bc.plan.flops * replicas
    line = {
        "config": name, "replicas": replicas, "steps": bc.plan.n_steps,
        "ms_per_pass": round(wall * 1e3, 3), "contractions_per_s": round(replicas / wall, 2),
        "flop_per_contraction": bc.plan.flops, "tflops": round(flops / wall / 1e12, 2),
        "bytes_min": bc.plan.bytes_min,
        "kernels": {k: {"steps": v["steps"], "ms": round(v["ms"], 3),
                        "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["ms"] > 0 else None}
                    for k, v in by.items()},
    }
    tiles = {}
    for t in bc.executor.step_tiles():
        tiles[str(t)] = tiles.get(str(t), 0) + 1
    line["tiles"] = tiles
    if extra:
        line.update(extra)
    print(json.dumps(line), flush=True)


def cfg3b(batch=4096, n_sites=100, bond=256, phys=4):
    tn = TN()
    hub = tn.add_copy_node(n_sites + 1)
    cores = [Shape((phys, bond) if i in (0, n_sites - 1) else (phys, bond, bond)) for i in range(n_sites)]
    nodes = nets.add_mps(tn, cores)
    for i, node in enumerate(nodes):
        inp = tn.add_input_node((batch, phys), var_shape_axes=(0,))
        tn.connect_nodes(inp, node, 1, 0)
        tn.connect_nodes(hub, inp, i, 0)
    shapes = [c.shape for c in cores] + [(batch, phys)] * n_sites
    path = ssa_to_linear(nets.batched_mps_path(n_sites), 2 * n_sites)
    bc, ops, out, wall, ms, by = run_device(tn.einsum_str, shapes, path, replicas=1, scale=16.0 ** 0.5)
    summarize(f"3b batched MPS B={batch} sites={n_sites} D={bond}", bc, wall, by, 1)


def cfg3a(sites=100, bond=256, phys=4):
    """The metric's network at the replica counts SURVEY.md 8(d) names (bench.py runs it at R = 512)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    tn, einstr, shapes, path = bench.build_network(sites, bond, phys)
    for R in (1, 8, 64):
        bc, ops, out, wall, ms, by = run_device(einstr, shapes, path, replicas=R, iters=10, scale=16.0)
        summarize(f"3a MPS-{sites} overlap D={bond} d={phys} zipper", bc, wall, by, R)
        del ops, out, bc
        torch.cuda.empty_cache()


def cfg4():
    for name, builder in (
        ("4(i) CP-hyper r=1024 n=1024", lambda: ("ac,ad,ae->cde", [(1024, 1024)] * 3)),
        ("4(ii) Tucker dense hub 1024^3", lambda: ("abc,ae,bf,cg->efg", [(1024, 1024, 1024)] + [(1024, 1024)] * 3)),
        # 4(iv) CP-wide r=4096: the Khatri-Rao intermediate has 2^32 elements (> the engine's 2^31 limit)
    ):
        einstr, shapes = builder()
        terms, out_l, sizes = E.paths.parse_einsum_input(einstr, shapes)
        path = E.paths.find_path(terms, out_l, sizes, "optimal")
        bc, ops, out, wall, ms, by = run_device(einstr, shapes, tuple(path), replicas=1, iters=3, scale=32.0)
        # spot-check a few entries of the 4 GiB result against the definition
        o = out[0]
        if einstr.startswith("ac"):
            A, B, C = [t.double() for t in ops[0]]
            c, d, e = 5, 700, 1023
            ref = float((A[:, c] * B[:, d] * C[:, e]).sum())
        else:
            H, A, B, C = [t for t in ops[0]]
            e, f, g = 5, 700, 1023
            ref = float(torch.einsum("abc,a,b,c->", H.double(), A[:, e].double(), B[:, f].double(), C[:, g].double()))
            c, d, e = e, f, g
        log_scale = bc.fetch_log_scale()[0]
        got = float(o[c, d, e]) * float(np.exp(log_scale))
        summarize(name, bc, wall, by, 1, {"spot_check_rel_err": abs(got - ref) / max(abs(ref), 1e-30)})
        del ops, out, bc
        torch.cuda.empty_cache()


def cfg5(rows=8, cols=8, bond=8):
    tn = nets.peps_closed(TN, rows, cols, 2, dtype=np.float32, seed=6)  # graph only (bond 2 placeholders)
    shapes = []
    for p in tn.params:
        shapes.append(tuple(bond if (d == 2 and p.ndim > 1 and ax > 0) else d for ax, d in enumerate(p.shape)))
    path = ssa_to_linear(nets.peps_row_path(rows, cols), 2 * rows * cols)
    bc, ops, out, wall, ms, by = run_device(tn.einsum_str, shapes, path, replicas=1, iters=3, scale=bond ** 0.5)
    summarize(f"5 PEPS {rows}x{cols} D={bond} row sweep (unsliced)", bc, wall, by, 1,
              {"largest_intermediate": max(i["out_numel"] for i in bc.plan.step_infos())})
    del ops, out, bc
    torch.cuda.empty_cache()
    # the same network on the path the library finds by itself (optimize="auto": noisy greedy + subtree reconfiguration)
    t0 = time.perf_counter()
    terms, out_l, sizes = E.paths.parse_einsum_input(tn.einsum_str, shapes)
    auto = tuple(E.paths.find_path(terms, out_l, sizes, "auto"))
    search_s = time.perf_counter() - t0
    bc, ops, out, wall, ms, by = run_device(tn.einsum_str, shapes, auto, replicas=1, iters=3, scale=bond ** 0.5)
    summarize(f"5 PEPS {rows}x{cols} D={bond} optimize='auto' path", bc, wall, by, 1,
              {"largest_intermediate": max(i["out_numel"] for i in bc.plan.step_infos()), "path_search_s": round(search_s, 2)})


def cfg5_sliced(rows=8, cols=8, bond=8, n_bonds=3):
    """Same network, sliced over n_bonds vertical bonds in the middle: bond**n_bonds slices run as
    replicas of one plan (zero-copy pointer offsets) - the single-GPU leg of the multi-GPU scheme."""
    from contractn_amd import dist

    rng = np.random.default_rng(6)
    tn = nets.peps_closed(TN, rows, cols, 2, dtype=np.float32, seed=6)
    shapes = [tuple(bond if (d == 2 and p.ndim > 1 and ax > 0) else d for ax, d in enumerate(p.shape)) for p in tn.params]
    ops = [(rng.standard_normal(s) / bond ** 0.5).astype(np.float32) for s in shapes]
    path = ssa_to_linear(nets.peps_row_path(rows, cols), 2 * rows * cols)
    terms = tn.einsum_str.split("->")[0].split(",")
    mid = rows // 2
    labels = tuple(next(iter(set(terms[(mid - 1) * cols + k]) & set(terms[mid * cols + k])))
                   for k in range(2, 2 + n_bonds))
    sc = dist.SlicedContraction(tn.einsum_str, ops, labels, optimize=path, rank=0, world=1)
    sc.run()
    t0 = time.perf_counter()
    iters = 3
    for _ in range(iters):
        t_s, c_s = sc.run()
    wall = (time.perf_counter() - t0) / iters
    full_plan = E._native_plan(E._contract_path(tn.einsum_str, tuple(shapes), optimize=path, memory_limit=None,
                                                use_blas=True), tuple(shapes), "float32")
    print(json.dumps({"config": f"5 PEPS {rows}x{cols} D={bond} sliced over {n_bonds} bonds", "slices": sc.n_total,
                      "ms_per_contraction": round(wall * 1e3, 3),
                      "flop_sliced_total": sc.bc.plan.flops * sc.n_total, "flop_unsliced": full_plan.flops,
                      "tflops_sliced": round(sc.bc.plan.flops * sc.n_total / wall / 1e12, 2),
                      "result": [float(t_s), float(c_s)]}), flush=True)
    # the unsliced answer for the same tensors
    import torch
    bc = E.BatchedContraction(tn.einsum_str, shapes, np.float32, optimize=path, replicas=1)
    outs, logs = bc.run_host([ops])
    print(json.dumps({"unsliced_result": [float(outs[0]), float(logs[0])]}), flush=True)


def cfg12():
    for name, build, dtype in (("1 copy node order 101", 1, np.float64), ("2 chain 1000x(3x3)", 2, np.float64)):
        tn = TN()
        if build == 1:
            hub = tn.add_copy_node(101)
            for i in range(100):
                tn.connect_nodes(hub, tn.add_dense_node(np.array([1, 0.99])), i, 0)
        else:
            prev = tn.add_dense_node(np.ones(3))
            for _ in range(1000):
                mat = tn.add_dense_node(np.ones((3, 3)))
                tn.connect_nodes(prev, mat, -1, 0)
                prev = mat
        fun = tn.make_contract_fun(split_format=True)
        params = tn.params
        fun(params, ())
        ts = []
        for _ in range(20):
            t0 = time.perf_counter()
            fun(params, ())
            ts.append(time.perf_counter() - t0)
        shapes = [p.shape for p in params]
        for R in (1, 256):
            bc = E.BatchedContraction(tn.einsum_str, shapes, dtype, replicas=R)
            sets = [list(params)] * R
            bc.run_host(sets)
            bc.executor.set_timing(3)
            for _ in range(3):
                bc.run_host(sets)
            dev_ms = float(bc.executor.step_ms().sum())
            print(json.dumps({"config": name, "replicas": R, "steps": bc.plan.n_steps,
                              "call_ms_median": round(float(np.median(ts)) * 1e3, 3) if R == 1 else None,
                              "device_ms_per_pass": round(dev_ms, 4),
                              "device_us_per_step": round(dev_ms * 1e3 / bc.plan.n_steps, 3),
                              "device_contractions_per_s": round(R / (dev_ms * 1e-3), 1)}), flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["cfg1", "cfg3a", "cfg3b", "cfg4", "cfg5"]
    if "cfg1" in which or "cfg2" in which:
        cfg12()
    if "cfg3a" in which:
        cfg3a()
    if "cfg3b" in which:
        cfg3b(batch=4096)
        cfg3b(batch=1024)
    if "cfg4" in which:
        cfg4()
    if "cfg5" in which:
        cfg5(8, 8, 8)
    if "cfg5s" in which:
        cfg5_sliced(8, 8, 8, 3)
