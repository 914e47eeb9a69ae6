#!/usr/bin/env python3
"""Development tool: build every stage PLAN of a staged sliced contraction on the host (no GPU) and list its steps.

    python tools/staged_plan_dry.py [--bond 16] [--max-intermediate 4294967296] [--stage K]

The executor, streams and device buffers of dist.StagedSlicedContraction are replaced by stubs; plans are the real ones."""
import argparse
import os
import sys
from unittest import mock

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from contractn_amd import dist, engine  # noqa: E402
from contractn_amd import einsum as E  # noqa: E402


class _Exec:
    def __init__(self, plan, replicas=1, device=0, stream=None):
        self.plan = plan

    def __getattr__(self, name):
        return lambda *a, **k: None


class _Stream:
    cuda_stream = 0

    def __init__(self, *a, **k):
        pass


class _Buf:
    def __init__(self, *a, **k):
        pass

    def data_ptr(self):
        return 1 << 20

    def __getitem__(self, i):
        return self


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bond", type=int, default=16)
    ap.add_argument("--max-intermediate", type=int, default=1 << 32)
    ap.add_argument("--slices", type=int, default=64)
    ap.add_argument("--stage", type=int, default=None)
    args = ap.parse_args()
    einstr, shapes, _ops = bench.peps_network(8, 8, args.bond)
    labels, path, rep = dist.staged_plan(einstr, shapes, min_slices=args.slices, max_intermediate=args.max_intermediate)
    ops = [np.broadcast_to(np.float32(0), s) for s in shapes]          # shapes only
    real_device = torch.device
    seen = []
    real_plan = E._native_plan

    def logging_plan(clist, shp, *a, **k):
        try:
            p_ = real_plan(clist, shp, *a, **k)
        except Exception as exc:  # noqa: BLE001
            print("PLAN FAILED:", exc)
            print("  shapes:", shp)
            for q, c in enumerate(clist):
                print("  step", q, c[0], c[2])
            raise
        seen.append(p_)
        return p_

    with mock.patch.object(engine, "Executor", _Exec), mock.patch.object(torch.cuda, "Stream", _Stream), \
            mock.patch.object(torch, "device", lambda *a: real_device("cpu")), mock.patch.object(torch, "zeros", _Buf), \
            mock.patch.object(torch, "as_tensor", _Buf), mock.patch.object(E, "_native_plan", logging_plan):
        try:
            sc = dist.StagedSlicedContraction(einstr, ops, labels, optimize=path, rank=0, world=1, device=0)
        except Exception as exc:  # noqa: BLE001
            import traceback
            print("constructor stopped:", repr(exc)[:300], traceback.format_exc().splitlines()[-4:])
            sc = None
    stages = sc.stages if sc is not None else [{"bc": type("B", (), {"plan": p_})(), "R": 0, "out_term": "?"} for p_ in seen[1::2]]
    for k, st in enumerate(stages):
        if args.stage is not None and k != args.stage:
            continue
        plan = st["bc"].plan
        print(f"stage {k}: R={st['R']} steps={plan.n_steps} out={st['out_term']}")
        for q, i in enumerate(plan.step_infos()):
            if i["flops"] * 1e-9 < 1.0:
                continue
            print(f"   step {q:3d} kernel={i['kernel']} swapped={i['swapped']} batch={i['batch']} m={i['m']} n={i['n']} k={i['k']} "
                  f"modes=({i['mode_a']},{i['mode_b']}) tile=({i['tile_m']},{i['tile_n']}) GF={i['flops'] * 1e-9:.0f}")


if __name__ == "__main__":
    main()
