"""Multi-process path (gloo, world_size 2, CPU): index slicing + the single all_gather join.

The per-slice compute is injected (the CPU oracle) so this runs without a GPU; on the GPU the
same code path uses the HIP engine and RCCL.
"""
import os
import socket
import sys

import numpy as np
import pytest

from contractn_amd import dist as D
from oracle import cpu_ref
from tests.helpers import ROOT, load_golden


def oracle_contract(einstr, *ops, optimize=None, split_format=True):
    return cpu_ref.contract(einstr, *ops, path=None, split_format=split_format)


def test_shard_range_partitions():
    for n in (0, 1, 7, 64):
        for world in (1, 2, 3, 8):
            got = [i for r in range(world) for i in D.shard_range(n, r, world)]
            assert got == list(range(n))
            sizes = [len(D.shard_range(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def test_combine_split_is_a_sum():
    rng = np.random.default_rng(0)
    parts, total = [], 0
    for c in (700.0, 695.0, -20.0):
        t = rng.standard_normal((3, 2))
        parts.append((t, c))
        total = total + t * np.exp(c - 700.0)
    t, c = D.combine_split(parts)
    np.testing.assert_allclose(t * np.exp(float(c) - 700.0), total, rtol=1e-12)
    assert np.isclose(np.mean(np.abs(t)), 1.0)


def test_slicing_reproduces_full_contraction_single_process():
    g = load_golden("peps3x3_D2_f64")
    einstr = g["einsum_str"]
    labels = tuple(sorted(set(einstr.split("->")[0].replace(",", "")))[2:4])
    t, c = D.contract_sliced(einstr, g["operands"], labels, contract_fn=oracle_contract, rank=0, world=1)
    full = float(g["t_hat"]) * np.exp(float(g["log_scale"]))
    assert abs(float(t) * np.exp(float(c)) - full) <= 1e-10 * abs(full)


def _worker(rank, world, port, name, labels, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = load_golden(name)
        t, c = D.contract_sliced(g["einsum_str"], g["operands"], labels, contract_fn=oracle_contract)
        q.put((rank, np.asarray(t), float(c)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,nlab", [("peps3x3_D2_f64", 2), ("mps_open_random_f64", 1)])
def test_world_size_2_gloo_join(name, nlab):
    import torch.multiprocessing as mp

    g = load_golden(name)
    lhs, out = g["einsum_str"].split("->")
    cand = [s for s in sorted(set(lhs.replace(",", ""))) if s not in out]
    labels = tuple(cand[:nlab])
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, name, labels, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full = np.asarray(g["t_hat"], dtype=np.float64) * np.exp(float(g["log_scale"]))
    for _rank, t, c in results:
        np.testing.assert_allclose(t * np.exp(c), full, rtol=1e-10)
    # every rank holds the same joined result
    np.testing.assert_array_equal(results[0][1], results[1][1])
    assert results[0][2] == results[1][2]


def test_choose_slices_bounds_memory_and_feeds_the_sliced_contraction():
    """Automatic slice selection: bounds the largest intermediate / provides enough slices, and the labels it
    returns give the same value through the sliced path as the unsliced contraction."""
    from contractn_amd import TN, dist
    from contractn_amd.paths import ssa_to_linear
    from tests import networks as nets

    tn = nets.peps_closed(TN, 3, 3, 3, dtype=np.float64, seed=6)
    ops = list(tn.params)
    shapes = [o.shape for o in ops]
    path = ssa_to_linear(nets.peps_row_path(3, 3), 18)
    labels, rep = dist.choose_slices(tn.einsum_str, shapes, optimize=path, max_intermediate=9, min_slices=8)
    assert rep["largest_intermediate"] <= 9 < rep["unsliced_largest_intermediate"]
    assert rep["slices"] >= 8 and rep["work_overhead"] >= 1.0
    assert all(lab not in tn.einsum_str.split("->")[1] for lab in labels)

    t_s, c_s = dist.contract_sliced(tn.einsum_str, ops, labels, contract_fn=oracle_contract)
    t_u, c_u = oracle_contract(tn.einsum_str, *ops)
    np.testing.assert_allclose(np.asarray(t_s) * np.exp(float(c_s)), np.asarray(t_u) * np.exp(float(c_u)), rtol=1e-10)


def test_choose_slices_with_path_co_optimises_and_reproduces_the_value():
    """Slice labels chosen together with the path: enough slices, bounded overhead, a smaller peak than a fixed
    path gives, and the sliced contraction on the returned path equals the unsliced one."""
    from contractn_amd import TN, dist
    from contractn_amd.paths import ssa_to_linear
    from tests import networks as nets

    tn = nets.peps_closed(TN, 4, 4, 3, dtype=np.float64, seed=9)
    ops = list(tn.params)
    shapes = [o.shape for o in ops]
    labels, path, rep = dist.choose_slices_with_path(tn.einsum_str, shapes, min_slices=8, trials=2)
    assert rep["slices"] >= 8 and len(labels) >= 2
    assert rep["largest_intermediate"] <= rep["unsliced_largest_intermediate"]
    fixed_labels, fixed = dist.choose_slices(tn.einsum_str, shapes, optimize=ssa_to_linear(nets.peps_row_path(4, 4), 32),
                                             min_slices=8)
    assert rep["work_overhead"] <= fixed["work_overhead"] * 1.5 + 1.0     # never far above the fixed-path choice
    t_s, c_s = dist.contract_sliced(tn.einsum_str, ops, labels, contract_fn=lambda e, *o, **kw: cpu_ref.contract(
        e, *o, path=list(path), split_format=kw.get("split_format", True)))
    t_u, c_u = oracle_contract(tn.einsum_str, *ops)
    np.testing.assert_allclose(np.asarray(t_s) * np.exp(float(c_s)), np.asarray(t_u) * np.exp(float(c_u)), rtol=1e-9)


def test_sliced_plan_is_cached_on_disk_and_validated(tmp_path):
    """`dist.sliced_plan`: the tens-of-seconds search runs once per network; the cached entry is re-used only
    for the same network and only when it is still a valid (labels, path) pair for it."""
    import json
    import time

    from contractn_amd import TN, dist
    from tests import networks as nets

    tn = nets.peps_closed(TN, 3, 3, 3, dtype=np.float64, seed=6)
    shapes = [o.shape for o in tn.params]
    first = dist.sliced_plan(tn.einsum_str, shapes, min_slices=9, cache_dir=str(tmp_path), trials=1)
    files = list(tmp_path.glob("sliced_*.json"))
    assert len(files) == 1 and first[2]["slices"] >= 9
    t0 = time.perf_counter()
    again = dist.sliced_plan(tn.einsum_str, shapes, min_slices=9, cache_dir=str(tmp_path), trials=1)
    assert time.perf_counter() - t0 < 0.05 and again[:2] == first[:2]
    # a corrupted entry (labels that are not in the network) is ignored and searched again
    d = json.load(open(files[0]))
    d["labels"] = ["☃"]
    json.dump(d, open(files[0], "w"))
    third = dist.sliced_plan(tn.einsum_str, shapes, min_slices=9, cache_dir=str(tmp_path), trials=1)
    assert third[:2] == first[:2]
    # the labels/path it returns contract to the unsliced value
    ops = list(tn.params)
    t_s, c_s = dist.contract_sliced(tn.einsum_str, ops, first[0], contract_fn=lambda e, *o, **kw: cpu_ref.contract(
        e, *o, path=list(first[1]), split_format=True))
    t_u, c_u = oracle_contract(tn.einsum_str, *ops)
    np.testing.assert_allclose(np.asarray(t_s) * np.exp(float(c_s)), np.asarray(t_u) * np.exp(float(c_u)), rtol=1e-10)


# ---- independent subtrees ------------------------------------------------------------------------------------------
def _oracle_on_path(einstr, *ops, optimize=None, split_format=True):
    path = None if isinstance(optimize, str) or optimize is None else list(optimize)
    return cpu_ref.contract(einstr, *ops, path=path, split_format=split_format)


def test_subtree_plan_cuts_a_grid_into_balanced_parts_with_small_boundaries():
    from contractn_amd import TN
    from tests import networks as nets

    tn = nets.peps_closed(TN, 4, 6, 3, dtype=np.float64, seed=6)
    shapes = [o.shape for o in tn.params]
    parts, top = D.subtree_plan(tn.einsum_str, shapes, n_parts=2)
    assert len(parts) == 2 and sorted(i for p in parts for i in p["operands"]) == list(range(48))
    assert all(len(p["out"]) <= 6 for p in parts)                      # at most one grid cut (+ nothing open)
    assert max(p["cost"] for p in parts) <= 3 * min(p["cost"] for p in parts)
    assert top["einsum"].endswith("->") and len(top["path"]) == 1
    # an explicit path is cut along ITS tree: every part is a subtree of it, the top part joins them
    from contractn_amd.paths import ssa_to_linear

    row = ssa_to_linear(nets.peps_row_path(4, 6), 48)
    parts_r, top_r = D.subtree_plan(tn.einsum_str, shapes, optimize=row, n_parts=2, max_boundary=3 ** 7)
    assert sorted(i for p in parts_r for i in p["operands"]) == list(range(48))
    t, c = D.contract_subtrees(tn.einsum_str, list(tn.params), optimize=row, contract_fn=_oracle_on_path,
                               rank=0, world=1, n_parts=2, max_boundary=3 ** 7)
    t_u, c_u = cpu_ref.contract(tn.einsum_str, *tn.params, path=row, split_format=True)
    assert float(t) == float(t_u) and abs(float(c) - float(c_u)) <= 1e-12 * max(1.0, abs(float(c_u)))


def test_assign_parts_balances_costs():
    owner = D.assign_parts([10, 9, 3, 3, 2, 1], 2)
    loads = [sum(c for c, o in zip([10, 9, 3, 3, 2, 1], owner) if o == r) for r in range(2)]
    assert sorted(loads) == [14, 14]


def _subtree_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    from contractn_amd import TN
    from tests import networks as nets

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tn = nets.peps_closed(TN, 4, 6, 3, dtype=np.float64, seed=6)
        t, c = D.contract_subtrees(tn.einsum_str, list(tn.params), contract_fn=_oracle_on_path)
        q.put((rank, float(t), float(c)))
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo_subtree_join():
    """Two ranks, each contracts the subtrees it owns, ONE all_gather of the packed results, top part on both."""
    import torch.multiprocessing as mp

    from contractn_amd import TN
    from contractn_amd.paths import ssa_to_linear
    from tests import networks as nets

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_subtree_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    tn = nets.peps_closed(TN, 4, 6, 3, dtype=np.float64, seed=6)
    t_u, c_u = cpu_ref.contract(tn.einsum_str, *tn.params, path=ssa_to_linear(nets.peps_row_path(4, 6), 48), split_format=True)
    assert results[0][1:] == results[1][1:]
    assert results[0][1] == float(t_u) and abs(results[0][2] - float(c_u)) <= 1e-11


# ---- batch sharding (data-parallel over a batch hyperedge) ---------------------------------------------------------
def _batched_case(batch=37, n_sites=5, bond=6, phys=3, dtype=np.float64):
    from contractn_amd import TN
    from contractn_amd import einsum as E
    from contractn_amd.paths import ssa_to_linear
    from tests import networks as nets

    tn, inputs = nets.batched_mps(TN, n_sites, bond, phys, batch, dtype=dtype, seed=4)
    ops = E.make_arg_packer(tn)(tn.params, inputs)
    path = ssa_to_linear(nets.batched_mps_path(n_sites), 2 * n_sites)
    label = tn.einsum_str.split("->")[1]
    assert len(label) == 1                      # the batch hyperedge is the only output label
    return tn.einsum_str, [np.asarray(o) for o in ops], path, label


def test_concat_split_and_single_process_batch_shards():
    einstr, ops, path, label = _batched_case()
    t_u, c_u = cpu_ref.contract(einstr, *ops, path=list(path), split_format=True)
    full = np.asarray(t_u, dtype=np.float64) * np.exp(float(c_u))
    # three emulated ranks: uneven chunks (13, 12, 12), joined by concat_split
    parts, covered = [], []
    for r in range(3):
        mine, lo, hi, axis = D.shard_batch_label(einstr, ops, label, r, 3)
        covered.append((lo, hi))
        assert axis == 0 and all(m.shape[0] == hi - lo for m, o in zip(mine, ops) if o.shape[0] == 37 and m.ndim == 2 and o.shape[1] == 3)
        parts.append(_oracle_on_path(einstr, *mine, optimize=path))
    assert covered == [(0, 13), (13, 25), (25, 37)]
    t, c = D.concat_split(parts, 0)
    np.testing.assert_allclose(t * np.exp(float(c)), full, rtol=1e-12)
    assert abs(np.mean(np.abs(t)) - 1.0) < 1e-12            # re-stabilised like a contraction step
    # world = 1 short-cut of the collective entry point
    t1, c1 = D.contract_batch_sharded(einstr, ops, label, optimize=path, contract_fn=_oracle_on_path, rank=0, world=1)
    np.testing.assert_allclose(t1 * np.exp(float(c1)), full, rtol=1e-12)
    with pytest.raises(ValueError):
        D.shard_batch_label(einstr, ops, "a" if label != "a" else "b", 0, 2)   # a summed label is not a batch


def _batch_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        einstr, ops, path, label = _batched_case()
        t, c = D.contract_batch_sharded(einstr, ops, label, optimize=path, contract_fn=_oracle_on_path)
        q.put((rank, np.asarray(t), float(c)))
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo_batch_shards_concatenate():
    """Two ranks, 19 + 18 of the 37 inputs each (padded all_gather), every rank ends with the full stabilised result."""
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_batch_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(((r, t, c) for r, t, c in (q.get(timeout=180) for _ in procs)), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    einstr, ops, path, _label = _batched_case()
    t_u, c_u = cpu_ref.contract(einstr, *ops, path=list(path), split_format=True)
    full = np.asarray(t_u, dtype=np.float64) * np.exp(float(c_u))
    for _r, t, c in results:
        assert t.shape == (37,)
        np.testing.assert_allclose(t * np.exp(c), full, rtol=1e-12)
    np.testing.assert_array_equal(results[0][1], results[1][1])
    assert results[0][2] == results[1][2]


def test_bench_gpus_n_starts_a_child_launcher(monkeypatch):
    """`python bench.py --gpus 4` outside a launcher spawns `python -m torch.distributed.run --nproc-per-node 4
    bench.py <same arguments>` as a CHILD process before anything touches the GPU, and exits with its code;
    under a launcher (WORLD_SIZE set) it does not."""
    import importlib
    import subprocess

    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    calls = []

    class Done:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        calls.append((cmd, env))
        return Done()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2", "--warmup", "1"])
    with pytest.raises(SystemExit) as exc:
        bench.main()
    assert exc.value.code == 7 and len(calls) == 1
    cmd, env = calls[0]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "2", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_batch_sharding_with_more_ranks_than_items_raises_on_every_rank():
    """A batch label shorter than the world size is a configuration error on EVERY rank (round-2 advice: only the
    empty ranks raised, the others hung in the all_gather)."""
    ops = [np.ones((2, 3)), np.ones((3,))]
    for rank in range(4):
        with pytest.raises(ValueError):
            D.contract_batch_sharded("bk,k->b", ops, "b", contract_fn=oracle_contract, rank=rank, world=4)


def _ascii(eq):
    table, res = {}, []
    for ch in eq:
        if ch in ",->":
            res.append(ch)
        else:
            table.setdefault(ch, "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ"[len(table)])
            res.append(table[ch])
    return "".join(res)


@pytest.mark.parametrize("min_saved", [0, 1 << 10, 1 << 60])
def test_stage_decomposition_evaluates_every_stage_once_per_value_of_its_labels(min_saved):
    """`dist.stage_decomposition` emulated with NumPy: every stage is evaluated once per joint value of the sliced
    labels IT depends on (cached here by exactly that key), the root once per slice, and the slices' sum equals the
    unsliced contraction.  ``min_saved = 0`` hoists every slice-independent subtree, a huge value none (one stage =
    plain slicing); the count of evaluations shrinks accordingly."""
    import itertools

    from contractn_amd import TN, paths
    from tests import networks as nets

    tn = nets.peps_closed(TN, 4, 4, 3, dtype=np.float64, seed=6)
    ops = list(tn.params)
    shapes = [o.shape for o in ops]
    labels, path, rep = D.choose_staged_slices(tn.einsum_str, shapes, min_slices=9, seeds=2)
    assert rep["slices"] >= 9 and rep["staged"] and rep["work_overhead"] <= rep["plain_overhead"]
    terms, _out, sizes = paths.parse_einsum_input(tn.einsum_str, shapes)
    stages = D.stage_decomposition(tn.einsum_str, shapes, labels, path, min_saved=min_saved)
    assert stages[-1]["dep"] == tuple(labels) and stages[-1]["out"] == ""
    cache = [dict() for _ in stages]

    def evaluate(k, assign):
        st = stages[k]
        key = tuple(assign[lab] for lab in st["dep"])
        if key not in cache[k]:
            arrs = []
            for kind, i in st["operands"]:
                if kind == "in":
                    arrs.append(ops[i][tuple(assign[lab] if lab in labels else slice(None) for lab in terms[i])])
                else:
                    assert i < k and set(stages[i]["dep"]) < set(st["dep"])      # lower stages first, fewer labels
                    arrs.append(evaluate(i, assign))
            clist = cpu_ref.contraction_list(st["einsum"], [a.shape for a in arrs], st["path"])
            t, c, _ = cpu_ref.core_contract(arrs, clist)
            cache[k][key] = np.asarray(t) * np.exp(float(c))
        return cache[k][key]

    total = 0.0
    for vals in itertools.product(*[range(sizes[lab]) for lab in labels]):
        total = total + evaluate(len(stages) - 1, dict(zip(labels, vals)))
    ref = np.einsum(_ascii(tn.einsum_str), *ops, optimize=True)
    assert abs(float(total) - float(ref)) <= 1e-10 * abs(float(ref))
    evals = sum(len(c) for c in cache)
    if min_saved == 0:
        assert len(stages) > 1 and any(st["dep"] == () for st in stages)
        assert all(len(c) == int(np.prod([sizes[lab] for lab in st["dep"]])) for c, st in zip(cache, stages))
    if min_saved == 1 << 60:
        assert len(stages) == 1 and evals == rep["slices"]


def test_hoisted_cost_counts_evaluations_per_dependency_set():
    """`paths.hoisted_cost`: a label contracted at the root slices for free; slicing a label of a corner repeats
    nothing either but leaves nothing to share out, which ``parallel`` charges for."""
    from contractn_amd import paths

    #   a - b - c      chain ab,bc,cd,de: contract (ab,bc) and (cd,de) first, then the halves over c
    terms = ["ab", "bc", "cd", "de"]
    sizes = {"a": 4, "b": 8, "c": 16, "d": 8, "e": 4}
    sets = [set(t) for t in terms]
    path = [(0, 1), (0, 1), (0, 1)]            # (ab,bc)->ac ; (cd,de)->ce ; (ac,ce)->ae
    base, _big = paths.path_cost(sets, "ae", sizes, path)
    tot, big, one = paths.hoisted_cost(sets, "ae", sizes, path, ["c"])
    assert tot == base and one * 16 == base          # c is summed at the root: every node carries it, no overhead
    tot_b, _b, one_b = paths.hoisted_cost(sets, "ae", sizes, path, ["b"])
    # b lives in the left half only: the right half (cd,de) is evaluated once, the rest 8 times at 1/8 of the size
    assert tot_b < one_b * 8 and tot_b == base + (8 - 1) * 4 * 16 * 4
    assert paths.hoisted_cost(sets, "ae", sizes, path, ["b"], parallel=8)[0] == one_b * 8


def test_bond16_plan_shards_as_documented():
    """The shipped 256-slice plan of the 8 x 8, D = 16 grid (two sliced labels, tensors of up to 2^32 elements): no work
    beyond the unsliced tree's on one rank, and the busiest rank's modelled work at 2 / 4 / 8 ranks under the rank grid the
    staged contraction picks (at 8 ranks the grid is 1 x 8 and 36 + 35 of a rank's 124 TFLOP are
    stages every rank computes in full)."""
    import bench
    from contractn_amd import paths

    einstr, shapes, _ops = bench.peps_network(8, 8, 16)
    shapes = [tuple(int(d) for d in s) for s in shapes]
    labels, path, rep = D.staged_plan(einstr, shapes, min_slices=64, max_intermediate=2 ** 32)
    assert rep["slices"] == 256 and len(labels) == 2 and rep["largest_intermediate"] == 2 ** 32
    assert abs(rep["work_overhead"] - 1.0) < 1e-3
    _terms, _out, sizes = paths.parse_einsum_input(einstr, shapes)
    sd = D.stage_decomposition(einstr, shapes, tuple(labels), path, min_saved=1 << 28)
    at = {lab: i for i, lab in enumerate(labels)}
    dep_idx = [[at[lab] for lab in st["dep"]] for st in sd]
    extents = [sizes[lab] for lab in labels]
    work = []
    for st in sd:
        lhs_ = st["einsum"].split("->")[0].split(",")
        work.append(paths.path_cost([set(t_) for t_ in lhs_], st["out"], sizes, st["path"])[0] if len(lhs_) > 1 else 0)

    def busiest(world):
        grid = D._best_rank_grid(extents, world, dep_idx, work)
        return grid, sum(w_ * int(np.prod([-(-extents[j] // grid[j]) for j in dep])) for dep, w_ in zip(dep_idx, work))

    g1, w1 = busiest(1)
    assert abs(w1 / rep["unsliced_flops"] - 1.0) < 1e-3
    g8, w8 = busiest(8)
    assert sorted(g8) == [1, 8]
    assert 3.9 < w1 / w8 < 4.1                    # the model's ceiling for 8 ranks (measured per-rank timing: 3.5 x)
    replicated = sum(w_ for dep, w_ in zip(dep_idx, work) if not dep) + min(
        sum(w_ * extents[dep[0]] for dep, w_ in zip(dep_idx, work) if dep == [j]) for j in range(2))
    assert 0.5 < replicated / w8 < 0.65           # what every rank computes in full: more than half of its work
