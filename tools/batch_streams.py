#!/usr/bin/env python3
"""Experiment: config 3b (batched MPS, B inputs through one 100-site D = 256 MPS) with the batch cut into S chunks,
every chunk its own executor on its own HIP stream (graph replays overlap on the card).

    python tools/batch_streams.py [B] [S ...]
"""
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
from contractn_amd.paths import ssa_to_linear  # noqa: E402
from tests import networks as nets  # noqa: E402
from tools.config_bench import Shape  # noqa: E402


def network(batch, n_sites=100, bond=256, phys=4):
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
    return tn.einsum_str, shapes, path


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    splits = [int(x) for x in sys.argv[2:]] or [1, 2, 4, 8]
    gen = torch.Generator(device="cuda")
    gen.manual_seed(0)
    for S in splits:
        einstr, shapes, path = network(B // S)
        bcs, launches, keep = [], [], []
        for s in range(S):
            bc = E.BatchedContraction(einstr, shapes, np.float32, optimize=path, replicas=1)
            ops = [torch.randn(sh, generator=gen, device="cuda") / 4.0 for sh in shapes]
            out = torch.zeros((1,) + tuple(bc.plan.out_shape), device="cuda")
            launches.append(bc.executor.make_enqueue([t.data_ptr() for t in ops], [out[0].data_ptr()]))
            bcs.append(bc); keep.append((ops, out))
        torch.cuda.synchronize()
        for _ in range(3):
            for l in launches:
                l()
        for bc in bcs:
            bc.executor.synchronize()
        iters = 10
        t0 = time.perf_counter()
        for _ in range(iters):
            for l in launches:
                l()
        for bc in bcs:
            bc.executor.synchronize()
        wall = (time.perf_counter() - t0) / iters
        flops = bcs[0].plan.flops * S
        print(json.dumps({"B": B, "chunks": S, "ms_per_pass": round(wall * 1e3, 3), "tflops": round(flops / wall / 1e12, 2),
                          "tiles": sorted(set(str(t) for t in bcs[0].executor.step_tiles()))}), flush=True)
        del bcs, launches, keep
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
