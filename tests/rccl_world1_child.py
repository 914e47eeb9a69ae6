"""Child of tests/test_gpu_dist.py::test_rccl_collectives_run_on_hardware_at_world_one, started under
`python -m torch.distributed.run --nproc-per-node 1` with CTN_JOIN_WORLD1=1: creates the `nccl` (= RCCL) process group
of ONE rank on cuda:0 and drives the joins of contractn_amd.dist through it - the packed all_gather of a closed sliced
network (`join_packed`) and reduce-scatter + all_reduce + all-gather of a large open result (`run_device`) - printing one
JSON line.  Not a test module (no `test_` prefix): it needs the launcher's environment."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import torch.distributed as tdist

    from contractn_amd import TN, dist
    from contractn_amd.paths import ssa_to_linear
    from tests import networks as nets

    torch.cuda.set_device(0)
    tdist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    out = {"backend": tdist.get_backend(), "world": tdist.get_world_size(), "join_alone": dist.join_alone()}
    try:
        # (1) closed network, plain and staged slicing: ONE all_gather of the packed (T_hat, c)
        rows = cols = 4
        tn = nets.peps_closed(TN, rows, cols, 4, dtype=np.float32, seed=6)
        ops = list(tn.params)
        row = ssa_to_linear(nets.peps_row_path(rows, cols), 2 * rows * cols)
        t_u, c_u = tn.contract(optimize=row, split_format=True)
        labels, path, _rep = dist.choose_staged_slices(tn.einsum_str, [o.shape for o in ops], min_slices=8, seeds=1)
        for name, make in (("plain", dist.SlicedContraction), ("staged", dist.StagedSlicedContraction)):
            sc = make(tn.einsum_str, ops, labels, optimize=path, rank=0, world=1, device=0)
            t, c = sc.run()
            out[name] = {"t": float(t), "c": float(c), "ok": bool(float(t) == float(t_u) and abs(float(c) - float(c_u)) <= 1e-3)}
        # (2) a large open result: reduce-scatter + all_reduce + all-gather on the device
        tn2 = nets.mps_open(TN, (8,) * 6, (4,) * 7, dtype=np.float32, seed=12)
        lhs, res = tn2.einsum_str.split("->")
        label = next(ch for ch in sorted(set(lhs.replace(",", ""))) if ch not in res)
        t2, c2 = tn2.contract(split_format=True)
        ref = t2.astype(np.float64) * np.exp(float(c2))
        sc2 = dist.SlicedContraction(tn2.einsum_str, list(tn2.params), (label,), optimize="auto", rank=0, world=1, device=0)
        td, cd = sc2.run_device()
        got = td.cpu().numpy().astype(np.float64) * np.exp(float(cd))
        out["open"] = {"numel": int(ref.size), "max_rel_err": float(np.max(np.abs(got - ref)) / np.max(np.abs(ref))),
                       "mean_abs_t_hat": float(np.mean(np.abs(td.cpu().numpy())))}
        out["open"]["ok"] = bool(out["open"]["max_rel_err"] <= 1e-4 and abs(out["open"]["mean_abs_t_hat"] - 1.0) < 1e-4)
    finally:
        tdist.destroy_process_group()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
