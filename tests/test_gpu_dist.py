"""World-size-2 runs whose compute is the HIP engine.

A GPU box of this pool has ONE card, so the two ranks are two fresh child processes that both compute on
``cuda:0`` and join over ``gloo`` (the rehearsal knobs ``CTN_BENCH_BACKEND=gloo CTN_BENCH_ONE_DEVICE=1`` of
bench.py); on a multi-GPU node the same code runs one rank per card over RCCL.  What is under test is the
product path end to end: slices dealt to the ranks, every rank running ITS slices as replicas of one plan on the
engine, ONE all_gather of ``(T_hat, c)`` and the log-sum-exp combine - against the unsliced engine result and
the CPU oracle."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from tests.helpers import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, rows, cols, bond, labels, path, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as tdist

    from contractn_amd import TN, dist, engine
    from tests import networks as nets

    tdist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert engine.device_count() >= 1
        tn = nets.peps_closed(TN, rows, cols, bond, dtype=np.float32, seed=6)
        sc = dist.SlicedContraction(tn.einsum_str, list(tn.params), labels, optimize=path, rank=rank, world=world,
                                    device=0)
        t, c = sc.run()                       # HIP engine on this rank's slices + the all_gather join
        q.put((rank, len(sc.my_slices), sc.n_total, float(t), float(c)))
    finally:
        tdist.destroy_process_group()


def test_two_ranks_hip_engine_and_all_gather_join():
    import torch.multiprocessing as mp

    from contractn_amd import TN, dist
    from contractn_amd.paths import ssa_to_linear
    from oracle import cpu_ref
    from tests import networks as nets

    rows = cols = 5
    bond = 4
    tn = nets.peps_closed(TN, rows, cols, bond, dtype=np.float32, seed=6)
    ops = list(tn.params)
    labels, path, rep = dist.choose_slices_with_path(tn.einsum_str, [o.shape for o in ops], min_slices=16, trials=1)
    assert rep["slices"] >= 16
    # references: the CPU oracle and the engine, both unsliced, on the row sweep
    row = ssa_to_linear(nets.peps_row_path(rows, cols), 2 * rows * cols)
    rt, rc = cpu_ref.contract(tn.einsum_str, *ops, path=row, split_format=True)
    et, ec = tn.contract(optimize=row, split_format=True)
    assert float(et) == float(rt) and abs(float(ec) - float(rc)) <= 1e-4

    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, rows, cols, bond, labels, path, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (r0, n0, tot0, t0, c0), (r1, n1, tot1, t1, c1) = results
    assert (r0, r1) == (0, 1) and tot0 == tot1 == rep["slices"] and n0 + n1 == tot0 and abs(n0 - n1) <= 1
    assert (t0, c0) == (t1, c1)                                  # every rank holds the same joined result
    assert t0 == float(rt) and abs(c0 - float(rc)) <= 1e-3       # north_star tolerance, fp32


def test_bench_peps_two_ranks_rehearsal():
    """`bench.py --config peps --gpus 2` under torch.distributed.run (both ranks on cuda:0, gloo): the JSON line
    carries the strong-scaling contract, the roofline of the dominant kernel and a finite, correct value."""
    env = dict(os.environ, CTN_BENCH_BACKEND="gloo", CTN_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--config", "peps", "--rows", "4", "--cols", "4", "--bond", "4", "--slices", "8", "--steps", "3",
              "--warmup", "2", "--cpu-seconds", "2"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common, env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    line1 = json.loads(one.stdout.strip().splitlines()[-1])
    assert line1["n_gpus"] == 1 and line1["scaling"] == "strong" and line1["config"]["slices"] >= 8
    assert line1["cpu_baseline"]["parity_vs_gpu"]["ok"], line1["cpu_baseline"]
    assert line1["unsliced_check"]["ok"], line1["unsliced_check"]
    assert line1["roofline"]["bound"] in ("mfma", "hbm") and line1["roofline"]["achieved"] > 0

    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                          os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common, env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert two.returncode == 0, two.stderr[-2000:]
    line2 = json.loads([ln for ln in two.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert line2["n_gpus"] == 2 and line2["config"]["slices_per_gpu"] * 2 == line2["config"]["slices"]
    # the same network, the same slices: the joined value does not depend on how many ranks shared the work
    assert line2["result"]["t_hat"] == line1["result"]["t_hat"]
    assert abs(line2["result"]["log_scale"] - line1["result"]["log_scale"]) <= 1e-5


def _subtree_rank_main(rank, world, port, rows, cols, bond, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as tdist

    from contractn_amd import TN, dist
    from tests import networks as nets

    tdist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tn = nets.peps_closed(TN, rows, cols, bond, dtype=np.float32, seed=6)
        t, c = dist.contract_subtrees(tn.einsum_str, list(tn.params), device=0)   # HIP engine on this rank's subtrees
        q.put((rank, float(t), float(c)))
    finally:
        tdist.destroy_process_group()


def test_two_ranks_independent_subtrees_with_one_all_gather():
    """north_star's second sharding mode: the 8 x 8 PEPS (D = 3: a cut has at most 3^12 elements) is split into
    balanced groups of tensors, each rank contracts its groups on the engine, the small results cross in ONE
    all_gather and both ranks finish the top of the tree - equal to the oracle's row sweep within 1e-3."""
    import torch.multiprocessing as mp

    from contractn_amd import TN
    from contractn_amd.paths import ssa_to_linear
    from oracle import cpu_ref
    from tests import networks as nets

    rows = cols = 8
    tn = nets.peps_closed(TN, rows, cols, 3, dtype=np.float32, seed=6)
    rt, rc = cpu_ref.contract(tn.einsum_str, *tn.params, path=ssa_to_linear(nets.peps_row_path(rows, cols), 128),
                              split_format=True)
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_subtree_rank_main, args=(r, 2, port, rows, cols, 3, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert results[0][1:] == results[1][1:]
    assert results[0][1] == float(rt) and abs(results[0][2] - float(rc)) <= 1e-3


def _open_network():
    """An MPS with 7 open physical legs (4^7 = 16384 output elements) and bonds of 8."""
    from contractn_amd import TN
    from tests import networks as nets

    return nets.mps_open(TN, (8,) * 6, (4,) * 7, dtype=np.float32, seed=12)


def _open_rank_main(rank, world, port, label, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as tdist

    from contractn_amd import dist

    tdist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tn = _open_network()
        sc = dist.SlicedContraction(tn.einsum_str, list(tn.params), (label,), optimize="auto", rank=rank, world=world, device=0)
        t, c = sc.run_device()                     # HIP engine for the slices AND for the local split-format sum
        q.put((rank, t.cpu().numpy(), float(c)))
    finally:
        tdist.destroy_process_group()


def test_open_output_device_join_one_and_two_ranks():
    """A LARGE open result (16384 elements) never goes through the host: per rank the slices' results are summed
    in split format by the engine, across ranks the tensor is reduced and re-stabilised in pieces (all_reduce
    here over gloo; reduce-scatter + all-gather over RCCL).  Equal to the unsliced contraction element by element."""
    import torch.multiprocessing as mp

    from contractn_amd import dist

    tn = _open_network()
    ops = list(tn.params)
    lhs, out = tn.einsum_str.split("->")
    label = next(c for c in sorted(set(lhs.replace(",", ""))) if c not in out)      # a bond: 8 slices
    t_u, c_u = tn.contract(split_format=True)
    ref = t_u.astype(np.float64) * np.exp(float(c_u))
    assert ref.size == 4 ** 7 >= dist.DEVICE_JOIN_MIN_NUMEL
    # one rank: device combine only
    sc = dist.SlicedContraction(tn.einsum_str, ops, (label,), optimize="auto", rank=0, world=1)
    t1, c1 = sc.run()
    got1 = t1.astype(np.float64) * np.exp(float(c1))
    assert np.max(np.abs(got1 - ref)) <= 1e-4 * np.max(np.abs(ref))
    assert abs(np.mean(np.abs(t1)) - 1.0) < 1e-4
    # two ranks
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_open_rank_main, args=(r, 2, port, label, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted((r, t, c) for r, t, c in (q.get(timeout=300) for _ in procs))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for _r, t, c in results:
        got = t.astype(np.float64) * np.exp(c)
        assert np.max(np.abs(got - ref)) <= 1e-4 * np.max(np.abs(ref))
        assert abs(np.mean(np.abs(t)) - 1.0) < 1e-4
    assert np.array_equal(results[0][1], results[1][1]) and results[0][2] == results[1][2]


def _batch_rank_main(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as tdist

    from contractn_amd import dist

    tdist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        einstr, ops, path, label = _batched_network()
        t, c = dist.contract_batch_sharded(einstr, ops, label, optimize=path, device=0)   # HIP engine on this rank's chunk
        q.put((rank, np.asarray(t), float(c)))
    finally:
        tdist.destroy_process_group()


def _batched_network():
    """BASELINE config 3b in small: 1000 inputs through a 6-site MPS (D = 64, d = 4) hanging on a batch hyperedge."""
    from contractn_amd import TN
    from contractn_amd import einsum as E
    from contractn_amd.paths import ssa_to_linear
    from tests import networks as nets

    tn, inputs = nets.batched_mps(TN, 6, 64, 4, 1000, dtype=np.float32, seed=4)
    ops = [np.asarray(o) for o in E.make_arg_packer(tn)(tn.params, inputs)]
    path = ssa_to_linear(nets.batched_mps_path(6), 12)
    return tn.einsum_str, ops, path, tn.einsum_str.split("->")[1]


def test_two_ranks_batch_sharded_classifier_outputs():
    """The data-parallel mode of SURVEY.md 8(e) for batched workloads: each rank evaluates its 500 inputs on the
    engine (one launch per interior site, the physical leg summed in the GEMM's epilogue), ONE all_gather
    concatenates the split-format chunks - equal to the unsharded engine result and to the oracle."""
    import torch.multiprocessing as mp

    from contractn_amd import contract
    from oracle import cpu_ref

    einstr, ops, path, label = _batched_network()
    t_u, c_u = contract(einstr, *ops, optimize=path, split_format=True)
    rt, rc = cpu_ref.contract(einstr, *ops, path=list(path), split_format=True)
    ref = np.asarray(rt, dtype=np.float64) * np.exp(float(rc))
    one = t_u.astype(np.float64) * np.exp(float(c_u))
    assert np.max(np.abs(one - ref)) <= 1e-3 * np.max(np.abs(ref))
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_batch_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(((r, t, c) for r, t, c in (q.get(timeout=300) for _ in procs)), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for _r, t, c in results:
        got = t.astype(np.float64) * np.exp(c)
        assert t.shape == (1000,) and t.dtype == np.float32
        assert np.max(np.abs(got - ref)) <= 1e-3 * np.max(np.abs(ref))
        assert np.max(np.abs(got - one)) <= 1e-5 * np.max(np.abs(one))
        assert abs(np.mean(np.abs(t)) - 1.0) < 1e-5
    assert np.array_equal(results[0][1], results[1][1]) and results[0][2] == results[1][2]


def test_bench_default_two_ranks_rehearsal():
    """The driver's multi-GPU launch of the DEFAULT bench (`torch.distributed.run ... bench.py --gpus N`), rehearsed
    with two ranks on cuda:0 over gloo at a small size: the weak-scaling headline line plus both secondaries - the
    batched MPS (data-parallel over the batch) and the PEPS sharded over the ranks - must come back without an error."""
    env = dict(os.environ, CTN_BENCH_BACKEND="gloo", CTN_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--sites", "12", "--bond", "64", "--replicas", "8", "--peps-bonds", "8"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert line["config"]["replicas_per_gpu"] == 8
    b = line["batched_mps"]
    assert "error" not in b and b["scaling"] == "weak" and b["value"] > 0 and b["epilogue_summed_steps"] == 10
    p = line["peps_strong_scaling"]["D8"]
    assert "error" not in p and p["scaling"] == "strong" and p["config"]["slices_per_gpu"] * 2 == p["config"]["slices"]
    assert p["result"]["t_hat"] in (1.0, -1.0) and np.isfinite(p["result"]["log_scale"])


def test_bench_starts_its_own_ranks_and_reports_the_best_single_gpu():
    """`python bench.py --gpus 2 --config peps ...` with NO launcher in the command (round-2 verdict, item 1): the
    script starts its two ranks itself as a child `torch.distributed.run` (both on cuda:0 over gloo here), passes
    their one JSON line through and returns their exit code; the line quotes the strong-scaling figure against the
    best single-GPU form (unsliced) as well as against the sliced plan on one GPU, both checked against the
    sharded value."""
    env = dict(os.environ, CTN_BENCH_BACKEND="gloo", CTN_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "peps", "--rows", "4", "--cols", "4",
           "--bond", "4", "--slices", "8", "--steps", "3", "--warmup", "2", "--single-gpu-reference", "--cross-check"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert line["config"]["slices_per_gpu"] * 2 == line["config"]["slices"]
    assert line["cross_check"]["ok"] is True and line["cross_check"]["slices"] >= 8     # another slicing, the same value
    ss = line["strong_scaling"]
    assert ss["n_gpus"] == 2 and ss["sliced_single_gpu_agrees"] is True and ss["unsliced_single_gpu_agrees"] is True
    assert ss["best_single_gpu_ms"] == min(ss["sliced_single_gpu_ms"], ss["unsliced_single_gpu_ms"])
    assert ss["speedup_vs_best_single_gpu"] > 0 and ss["speedup_vs_sliced_single_gpu"] >= ss["speedup_vs_best_single_gpu"]


def test_combine_split_kernel_matches_the_host_combine():
    """`ctn_exec_combine_split` (the device-side join) against `dist.combine_split` (NumPy): live and exact-zero
    parts, a part whose scale is far below the maximum, fp32 and fp64 parts, scalar and small-tensor payloads;
    all parts zero gives (0, 0)."""
    import torch

    from contractn_amd import dist
    from contractn_amd import einsum as E

    bc = E.BatchedContraction("ab,bc->ac", [(4, 4), (4, 4)], np.float32, optimize=((0, 1),), replicas=1)
    ex = bc.executor
    rng = np.random.default_rng(5)
    for dtype, tdt in ((np.float32, torch.float32), (np.float64, torch.float64)):
        for numel, n in ((1, 8), (1, 300), (6, 5), (1000, 7)):
            t = rng.standard_normal((n, numel)).astype(dtype)
            c = rng.uniform(-30, 30, n)
            t[1] = 0                       # an exact zero whose scale would otherwise be the maximum
            c[1] = 500.0
            if n > 3:
                c[3] = -2000.0             # a live part that cannot contribute
            ref_t, ref_c = dist.combine_split([(t[i], c[i]) for i in range(n)])
            d_t, d_c = torch.as_tensor(t, device="cuda"), torch.as_tensor(c, device="cuda")
            out = torch.zeros(numel + 1, dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()
            ex.combine_split(d_t.data_ptr(), numel, d_c.data_ptr(), 1, n, numel, out.data_ptr(), dtype=dtype)
            ex.synchronize()
            got = out.cpu().numpy()
            assert abs(got[numel] - float(ref_c)) <= 1e-12 * max(1.0, abs(float(ref_c)))
            np.testing.assert_allclose(got[:numel], np.asarray(ref_t, dtype=np.float64).ravel(),
                                       rtol=1e-6 if dtype == np.float32 else 1e-12, atol=1e-12)
    zeros = torch.zeros(4, 3, device="cuda")
    out = torch.full((4,), 7.0, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    ex.combine_split(zeros.data_ptr(), 3, torch.ones(4, dtype=torch.float64, device="cuda").data_ptr(), 1, 4, 3, out.data_ptr())
    ex.synchronize()
    assert np.array_equal(out.cpu().numpy(), np.zeros(4))
    ex.close()


def test_device_join_equals_host_join_also_for_a_tiny_slice(monkeypatch):
    """The device join (`SlicedContraction.run` -> `run_small`) and the host join (`CTN_HOST_JOIN=1`: per-slice
    fetch + `combine_split`) agree on a sliced PEPS; and a slice whose result is tiny but not zero (sum |T| below
    the 1e-7 threshold: never rescaled) is kept by both - the liveness rule is "not an exact zero" (round-2
    advice), here on an open-output network through `local_result_device` as well."""
    from contractn_amd import TN, dist
    from tests import networks as nets

    tn = nets.peps_closed(TN, 4, 4, 4, dtype=np.float32, seed=6)
    ops = list(tn.params)
    labels, path, rep = dist.choose_slices_with_path(tn.einsum_str, [o.shape for o in ops], min_slices=8, trials=1)
    sc = dist.SlicedContraction(tn.einsum_str, ops, labels, optimize=path, workspace_budget=1 << 20)
    assert len(sc._chunks) >= 1
    t_d, c_d = sc.run()
    monkeypatch.setenv("CTN_HOST_JOIN", "1")
    t_h, c_h = sc.run()
    monkeypatch.delenv("CTN_HOST_JOIN")
    assert float(t_d) == float(t_h) and abs(float(c_d) - float(c_h)) <= 2e-5      # fp32: logf on the device vs NumPy
    # two slices of "ka,kb->ab": slice 0 is ~1e-10 in magnitude (below min_norm, not rescaled), slice 1 is an exact zero
    a = np.zeros((2, 3), dtype=np.float32)
    b = np.zeros((2, 5), dtype=np.float32)
    a[0] = [1e-5, -2e-5, 3e-5]
    b[0] = [1e-5, 2e-5, -1e-5, 3e-5, 1e-5]
    sc2 = dist.SlicedContraction("ka,kb->ab", [a, b], ("k",), optimize=((0, 1),))
    ref = np.einsum("ka,kb->ab", a.astype(np.float64), b.astype(np.float64))
    t2, c2 = sc2.run()
    np.testing.assert_allclose(np.asarray(t2, dtype=np.float64) * np.exp(float(c2)), ref, rtol=1e-5)
    t3, c3 = sc2.local_result_device()
    np.testing.assert_allclose(t3.cpu().numpy().astype(np.float64) * np.exp(float(c3)), ref, rtol=1e-5)


def test_staged_slicing_equals_plain_slicing_and_the_unsliced_value():
    """`StagedSlicedContraction`: slice-independent parts of the tree are contracted once, parts below one sliced
    label once per value of it, only the root stage once per slice - same value as plain slicing (every slice
    repeats everything) and as the unsliced network, slice by slice; fewer evaluations than plain slicing; two
    emulated ranks partition the root's slices and evaluate only what those project onto."""
    from contractn_amd import TN, dist
    from contractn_amd.paths import ssa_to_linear
    from tests import networks as nets

    rows = cols = 5
    tn = nets.peps_closed(TN, rows, cols, 4, dtype=np.float32, seed=6)
    ops = list(tn.params)
    shapes = [o.shape for o in ops]
    row = ssa_to_linear(nets.peps_row_path(rows, cols), 2 * rows * cols)
    t_u, c_u = tn.contract(optimize=row, split_format=True)
    labels, path, rep = dist.choose_staged_slices(tn.einsum_str, shapes, min_slices=16, seeds=2)
    assert rep["slices"] >= 16 and rep["work_overhead"] < rep["plain_overhead"]
    st = dist.StagedSlicedContraction(tn.einsum_str, ops, labels, optimize=path, min_saved=1 << 12)
    assert len(st.stages) > 1
    ev = st.evaluations()
    assert ev[-1][1] == rep["slices"] and any(n < rep["slices"] for _dep, n, _all in ev[:-1])
    for _ in range(4):                                   # eager launches, graph capture, replays: the same bits
        t_s, c_s = st.run()
        assert float(t_s) == float(t_u) and abs(float(c_s) - float(c_u)) <= 1e-3
    plain = dist.SlicedContraction(tn.einsum_str, ops, labels, optimize=path)
    t_p, c_p = plain.run()
    assert float(t_p) == float(t_s) and abs(float(c_p) - float(c_s)) <= 2e-5
    pt, pc = plain.slices_host()
    gt, gc = st.slices_host()
    assert np.array_equal(pt.ravel(), gt.ravel()) and np.max(np.abs(pc - gc)) <= 2e-5     # slice by slice
    # a tiny workspace budget: one evaluation per launch, several launches per stage
    small = dist.StagedSlicedContraction(tn.einsum_str, ops, labels, optimize=path, min_saved=1 << 12, workspace_budget=1)
    assert small.R == 1
    t_k, c_k = small.run()
    assert float(t_k) == float(t_s) and abs(float(c_k) - float(c_s)) <= 1e-6
    # the leading labels walked in a host loop (stage buffers hold one group's evaluations at a time; stages are
    # recomputed when the group needs other evaluations than the ones held): same value, never fewer evaluations
    for outer in range(1, len(labels) + 1):
        looped = dist.StagedSlicedContraction(tn.einsum_str, ops, labels, optimize=path, min_saved=1 << 12, outer=outer)
        assert looped.outer == outer and looped.n_groups > 1
        for _ in range(3):
            t_o, c_o = looped.run()
            assert float(t_o) == float(t_s) and abs(float(c_o) - float(c_s)) <= 1e-6
        assert all(a[1] >= b[1] for a, b in zip(looped.evaluations(), ev))
        assert max(S["out"].shape[0] for S in looped.stages[:-1]) <= max(S["out"].shape[0] for S in st.stages[:-1])
    tight = dist.StagedSlicedContraction(tn.einsum_str, ops, labels, optimize=path, min_saved=1 << 12, held_budget=1)
    assert tight.outer == len(labels)                  # nothing fits: every label walked on the host
    # the root takes sliced labels back as ordinary contracted labels (forced here; by default only when a large operand
    # of the root would otherwise be re-read): fewer root evaluations, each the sum of several slices, the same value
    back = dist.StagedSlicedContraction(tn.einsum_str, ops, labels, optimize=path, min_saved=1 << 12, unslice_min_numel=1)
    if back.unsliced:
        assert len(back.root_members) < rep["slices"] and sum(len(m) for m in back.root_members) == rep["slices"]
        for _ in range(3):
            t_b, c_b = back.run()
            assert float(t_b) == float(t_s) and abs(float(c_b) - float(c_s)) <= 1e-5
        bt, bc_ = back.slices_host()
        for q, grp in enumerate(back.root_members):     # every root evaluation = the split-format sum of its slices
            rt, rc = dist.combine_split([(pt[i], pc[i]) for i in grp])
            assert float(bt[q]) == float(rt) and abs(float(bc_[q]) - float(rc)) <= 2e-5
    assert st.unsliced == ()                            # tiny operands: not worth a pass by default
    # two emulated ranks: disjoint halves of the slices, lower stages only where their slices need them
    parts = []
    for rank in range(2):
        half = dist.StagedSlicedContraction(tn.einsum_str, ops, labels, optimize=path, rank=rank, world=2, min_saved=1 << 12)
        assert len(half.my_slices) * 2 == rep["slices"]
        half.world = 1                                   # no process group here: the local part only
        parts.append(half.run())
    t_j, c_j = dist.combine_split(parts)
    assert float(t_j) == float(t_s) and abs(float(c_j) - float(c_s)) <= 1e-5


def test_scale_bookkeeping_kernels_match_numpy():
    """`ctn_exec_add_scales` / `ctn_exec_merge_scales` (what `StagedSlicedContraction.run` does between its stages, on
    the device) against NumPy: registers of consumed evaluations added in order; groups of evaluations along the merged
    axes brought to the largest register among their LIVE members - liveness over the whole tensor (an evaluation whose
    first 64 elements are zero but which is not zero counts: round-3 advice), exact zeros untouched, an all-zero group
    gets register 0 - fp32 and fp64, vector body and scalar tail."""
    import torch

    from contractn_amd import einsum as E

    bc = E.BatchedContraction("ab,bc->ac", [(4, 4), (4, 4)], np.float32, optimize=((0, 1),), replicas=1)
    ex = bc.executor
    rng = np.random.default_rng(9)
    # -- add_scales: 0, 2 and 11 children (more than one launch)
    for n, n_kids in ((1, 0), (300, 2), (70, 11)):
        own = rng.standard_normal(n)
        kids = [(rng.standard_normal(5 + 3 * j), rng.integers(0, 5 + 3 * j, n)) for j in range(n_kids)]
        want = own.copy()
        for sc, ix in kids:
            want = want + sc[ix]
        d_own = torch.as_tensor(own, device="cuda")
        d_dst = torch.zeros(n, dtype=torch.float64, device="cuda")
        d_kids = [(torch.as_tensor(sc, device="cuda"), torch.as_tensor(ix, device="cuda", dtype=torch.int64)) for sc, ix in kids]
        torch.cuda.synchronize()
        ex.add_scales(d_dst.data_ptr(), d_own.data_ptr(), n, [(a.data_ptr(), b.data_ptr()) for a, b in d_kids])
        ex.synchronize()
        assert np.array_equal(d_dst.cpu().numpy(), want)
    # -- merge_scales
    for dtype, tdt in ((np.float32, torch.float32), (np.float64, torch.float64)):
        for grid, merged, numel in (((3, 4, 2), (1,), 1002), ((3, 4, 2), (0, 2), 5000), ((6,), (0,), 70), ((2, 5), (1,), 262144 + 4)):
            n = int(np.prod(grid))
            stride = (numel * np.dtype(dtype).itemsize + 15) // 16 * 16 // np.dtype(dtype).itemsize
            buf = np.zeros((n, stride), dtype=dtype)
            buf[:, :numel] = rng.standard_normal((n, numel)).astype(dtype)
            cum = rng.uniform(-40, 40, n)
            buf[1, :] = 0                                   # an exact zero with the largest register of its group
            cum[1] = 300.0
            buf[2, :numel - 1] = 0                          # zero but for its LAST element: live
            cum[2] = 45.0
            axes = tuple(merged)
            if len(grid) == 1:
                buf[:] = 0                                  # a group of zeros: register 0
            live = (buf[:, :numel] != 0).any(axis=1).reshape(grid)
            cg = cum.reshape(grid)
            top = np.where(live, cg, -np.inf).max(axis=axes, keepdims=True)
            top = np.where(np.isinf(top), 0.0, top)
            with np.errstate(over="ignore"):                # (dead members may sit above their group's top: not used)
                fac = np.exp(cg - top).reshape(n).astype(dtype)
            want = buf.copy()
            lv = live.reshape(n)
            want[lv] = want[lv] * fac[lv, None]
            want_cum = np.broadcast_to(top, grid).reshape(n)
            d_buf = torch.as_tensor(buf, device="cuda")
            d_cum = torch.as_tensor(cum, device="cuda")
            torch.cuda.synchronize()
            ex.merge_scales(d_buf.data_ptr(), stride, numel, d_cum.data_ptr(), n, grid, [q in axes for q in range(len(grid))], dtype=dtype)
            ex.synchronize()
            assert np.array_equal(d_cum.cpu().numpy(), want_cum)
            got = d_buf.cpu().numpy()[:, :numel]
            # (the device's exp and NumPy's may differ in the last bit of the factor)
            np.testing.assert_allclose(got, want[:, :numel], rtol=1e-6 if dtype == np.float32 else 1e-14, atol=0)
            same = ~lv | (fac == 1)                         # exact zeros and the group's top member: not touched at all
            assert np.array_equal(got[same], buf[same][:, :numel])
    ex.close()


def test_staged_fallback_covers_exactly_the_ranks_own_block():
    """Round-3 advice: when the range check of a staged run fails on SOME ranks only, those ranks repeat their part on the
    plain, checked path - which must cover exactly the BLOCK of the label grid the staged form dealt to the rank, not the
    contiguous range plain slicing would give it.  Four emulated ranks, the check forced to fail on two of them: the
    joined value is the unsliced one, and at least one of the forced ranks owns a block that is not a contiguous range."""
    from contractn_amd import TN, dist
    from contractn_amd.paths import ssa_to_linear
    from tests import networks as nets

    rows = cols = 5
    tn = nets.peps_closed(TN, rows, cols, 4, dtype=np.float32, seed=6)
    ops = list(tn.params)
    shapes = [o.shape for o in ops]
    row = ssa_to_linear(nets.peps_row_path(rows, cols), 2 * rows * cols)
    t_u, c_u = tn.contract(optimize=row, split_format=True)
    labels, path, rep = dist.choose_staged_slices(tn.einsum_str, shapes, min_slices=16, seeds=2)
    world, forced = 4, (1, 2)
    parts, differs = [], False
    import itertools

    for rank in range(world):
        st = dist.StagedSlicedContraction(tn.einsum_str, ops, labels, optimize=path, rank=rank, world=world, min_saved=1 << 12)
        st.world = 1                                     # no process group here: the local part only
        if rank in forced:
            for S in st.stages:
                S["bc"].executor.scales_suspect = lambda *a, **k: True
            all_slices = list(itertools.product(*[range(st.sizes[lab]) for lab in labels]))
            contiguous = [all_slices[i] for i in dist.shard_range(len(all_slices), rank, world)]
            differs = differs or sorted(contiguous) != sorted(st.my_slices)
        parts.append(st.run())
        if rank in forced:
            assert st._plain is not None and sorted(st._plain.my_slices) == sorted(st.my_slices)
        else:
            assert st._plain is None
    assert differs, "the test network no longer separates block and range partitions"
    t_j, c_j = dist.combine_split(parts)
    assert float(t_j) == float(t_u) and abs(float(c_j) - float(c_u)) <= 1e-3


def _one_rank_launcher_env():
    env = dict(os.environ, CTN_JOIN_WORLD1="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "CTN_BENCH_BACKEND", "CTN_BENCH_ONE_DEVICE"):
        env.pop(k, None)
    return env


def test_rccl_collectives_run_on_hardware_at_world_one():
    """Round-3 verdict, item 1a: RCCL has to have met this code on a GPU before the first 8-GPU run.  A one-rank `nccl`
    process group on cuda:0 (a launcher started before anything touches the GPU), CTN_JOIN_WORLD1=1: `join_packed`'s
    all_gather (plain and staged slicing) and `run_device`'s reduce-scatter + all_reduce + all-gather execute on RCCL and
    give the unsliced values."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "rccl_world1_child.py")]
    out = subprocess.run(cmd, env=_one_rank_launcher_env(), cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["backend"] == "nccl" and line["world"] == 1 and line["join_alone"] is True
    assert line["plain"]["ok"] and line["staged"]["ok"], line
    assert line["open"]["ok"] and line["open"]["numel"] == 4 ** 7, line


def test_bench_peps_under_a_one_rank_launcher_over_rccl():
    """`bench.py --config peps --bond 8` exactly as the driver starts a rank (`torch.distributed.run`), with ONE rank
    and the `nccl` backend: process-group creation, barrier, the timing all_reduce and the per-contraction all_gather of
    the join all run on RCCL; the line says so and its value agrees with the unsliced network."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--config", "peps", "--bond", "8",
           "--steps", "5", "--warmup", "2", "--cpu-seconds", "3"]
    out = subprocess.run(cmd, env=_one_rank_launcher_env(), cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    col = line["collectives"]
    assert col == {"process_group": True, "backend": "nccl", "world": 1, "join_all_gather_per_contraction": True}
    assert line["n_gpus"] == 1 and line["config"]["slices"] >= 64 and line["value"] > 0
    assert line["unsliced_check"]["ok"], line["unsliced_check"]
    assert line["cpu_baseline"]["parity_vs_gpu"]["ok"], line["cpu_baseline"]


def test_report_suspect_makes_an_executor_stay_eager_after_three_runs_in_a_row():
    """`ctn_exec_report_suspect` (round-3 advice): a caller that checks the range itself and never fetches - the staged
    sliced contraction - tells the executor the verdict once per run; a clean run resets the streak, the third suspect run
    in a row leaves the executor in the eager rescale mode for good (its runs are then never suspect), and the values stay
    the oracle's."""
    from contractn_amd import einsum as E
    from oracle import cpu_ref

    rng = np.random.default_rng(3)
    ops = [rng.standard_normal((96, 64)).astype(np.float32), rng.standard_normal((64, 80)).astype(np.float32),
           rng.standard_normal((80, 8)).astype(np.float32)]
    bc = E.BatchedContraction("ab,bc,cd->ad", [o.shape for o in ops], np.float32, optimize=((0, 1), (0, 1)), replicas=1)
    ex = bc.executor
    t0, c0 = bc.run_host([ops])
    assert ex.report_suspect(True) == 1 and ex.report_suspect(True) == 2 and ex.report_suspect(False) == 0
    assert ex.set_rescale_mode(0) == 0                       # still lazy
    assert [ex.report_suspect(True) for _ in range(3)] == [1, 2, 3]
    t1, c1 = bc.run_host([ops])                              # an eager run now
    assert ex.set_rescale_mode(0) == 1                       # ... the executor had switched itself
    rt, rc = cpu_ref.contract("ab,bc,cd->ad", *ops, path=[(0, 1), (0, 1)], split_format=True)
    for t, c in ((t0, c0), (t1, c1)):
        got = t[0].astype(np.float64) * np.exp(float(c[0]))
        ref = np.asarray(rt, dtype=np.float64) * np.exp(float(rc))
        assert np.max(np.abs(got - ref)) <= 2e-5 * np.max(np.abs(ref))
    ex.close()
