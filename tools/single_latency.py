#!/usr/bin/env python3
"""Development tool: ONE network in flight - wall time per contraction of the headline network (100-site MPS overlap,
D = 256, fp32) and of smaller bonds, replicas = 1, tensors resident on the device; per-kernel split from events."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

import bench  # noqa: E402
import config_bench as cb  # noqa: E402


def main():
    bonds = [int(x) for x in sys.argv[1:]] or [256, 64]
    for bond in bonds:
        tn, einstr, shapes, path = bench.build_network(100, bond, 4)
        for R in (1, 4):
            bc, ops, out, wall, ms, by = cb.run_device(einstr, shapes, path, replicas=R, iters=20, scale=16.0)
            tiles = bc.executor.step_tiles()
            kinds = {}
            for t in tiles:
                kinds[str(t)] = kinds.get(str(t), 0) + 1
            print(json.dumps({"network": f"MPS-100 overlap D={bond}", "replicas": R, "ms_per_contraction_wall": round(wall * 1e3, 4),
                              "us_per_step": round(wall * 1e6 / bc.plan.n_steps, 2), "event_ms_sum": round(float(ms.sum()), 4),
                              "tflops": round(bc.plan.flops * R / wall / 1e12, 2), "tiles": kinds}), flush=True)
            del ops, out, bc
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
