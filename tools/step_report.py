#!/usr/bin/env python3
"""Development tool: per-step report (shape, kernel, tile, time, rate) of one network on a given path.

    python tools/step_report.py peps 8 8 8 auto        # rows cols bond optimize
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import config_bench as cb  # noqa: E402
from contractn_amd import TN  # noqa: E402
from contractn_amd import einsum as E  # noqa: E402
from contractn_amd.engine import KERNEL_NAMES  # noqa: E402
from contractn_amd.paths import ssa_to_linear  # noqa: E402
from tests import networks as nets  # noqa: E402


def main():
    rows, cols, bond = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    opt = sys.argv[5] if len(sys.argv) > 5 else "auto"
    tn = nets.peps_closed(TN, rows, cols, 2, dtype=np.float32, seed=6)
    shapes = [tuple(bond if (d == 2 and p.ndim > 1 and ax > 0) else d for ax, d in enumerate(p.shape)) for p in tn.params]
    if opt == "row":
        path = ssa_to_linear(nets.peps_row_path(rows, cols), 2 * rows * cols)
    else:
        terms, out_l, sizes = E.paths.parse_einsum_input(tn.einsum_str, shapes)
        path = tuple(E.paths.find_path(terms, out_l, sizes, opt))
    bc, ops, out, wall, ms, by = cb.run_device(tn.einsum_str, shapes, path, replicas=1, iters=3, scale=bond ** 0.5)
    infos, tiles = bc.plan.step_infos(), bc.executor.step_tiles()
    order = sorted(range(len(infos)), key=lambda s: -ms[s])
    print(f"wall {wall*1e3:.3f} ms, {len(infos)} steps; slowest 14:")
    for s in order[:14]:
        i = infos[s]
        byts = 4 * (i["m"] * i["k"] + i["k"] * i["n"] + i["m"] * i["n"]) * max(i["batch"], 1)
        print(f"  step {s:3d} {KERNEL_NAMES[i['kernel']]:9s} b={i['batch']:<5d} m={i['m']:<8d} n={i['n']:<6d} k={i['k']:<6d} modes=({i['mode_a']},{i['mode_b']}) "
              f"tile={tiles[s]}  {ms[s]*1e3:8.1f} us  {i['flops']/ms[s]/1e9:7.1f} TFLOP/s  {byts/ms[s]/1e9:6.2f} TB/s")


if __name__ == "__main__":
    main()
