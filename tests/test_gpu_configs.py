"""BASELINE configs 3a/3b/4/5 at realistic sizes on the GPU: oracle comparisons where the CPU
finishes in seconds, size-independent properties (linearity, batch independence, CP == Tucker
with a materialised copy tensor, sliced == unsliced) beyond that."""
import numpy as np
import pytest

from contractn_amd import TN, contract, dist
from contractn_amd import einsum as E
from contractn_amd.paths import ssa_to_linear
from oracle import cpu_ref
from tests import networks as nets

pytestmark = pytest.mark.gpu


def full(t, c):
    return np.asarray(t, dtype=np.float64) * np.exp(float(c))


def test_cfg3a_mps_overlap_bond256_vs_oracle():
    """Headline shapes (D=256, d=4, fp32, zipper path) at 12 sites; tolerance 1e-3 (north_star)."""
    tn, ssa = nets.mps_overlap(TN, 12, 256, 4, dtype=np.float32, seed=3)
    path = ssa_to_linear(ssa, 24)
    t, c = tn.contract(optimize=path, split_format=True)
    rt, rc = cpu_ref.contract(tn.einsum_str, *tn.params, path=path, split_format=True)
    assert t.dtype == np.float32 and abs(float(t)) == 1.0 and float(t) == float(rt)
    # scalar overlap of random states = a sum with cancellation: both fp32 paths carry ~1e-5 rel. error
    assert abs(float(c) - float(rc)) <= 1e-4 * max(1.0, abs(float(rc)))
    infos = E._native_plan(E._contract_path(tn.einsum_str, tuple(p.shape for p in tn.params), optimize=path,
                                            memory_limit=None, use_blas=True),
                           tuple(p.shape for p in tn.params), "float32").step_infos()
    assert sum(i["kernel"] == 2 for i in infos) == 21  # all bulk steps are MFMA GEMMs, and the closing 256 x 4 x 256 step a masked tile


def test_cfg3a_linearity_of_the_log_register():
    """Scaling one core by 2^k shifts the register by exactly k ln 2 and leaves T_hat unchanged."""
    tn, ssa = nets.mps_overlap(TN, 8, 128, 4, dtype=np.float32, seed=5)
    path = ssa_to_linear(ssa, 16)
    params = list(tn.params)
    fun = tn.make_contract_fun(optimize=path, split_format=True)
    t0, c0 = fun(tuple(params), ())
    params[3] = params[3] * np.float32(8.0)
    t1, c1 = fun(tuple(params), ())
    assert float(t1) == float(t0)
    assert abs((float(c1) - float(c0)) - 3 * np.log(2.0)) < 1e-6


def test_cfg3b_batched_inputs_through_batch_hyperedge():
    B, n_sites, bond, phys = 1024, 10, 64, 4
    tn, inputs = nets.batched_mps(TN, n_sites, bond, phys, B, dtype=np.float32, seed=4)
    path = ssa_to_linear(nets.batched_mps_path(n_sites), 2 * n_sites)
    fun = tn.make_contract_fun(optimize=path, split_format=True)
    t, c = fun(tn.params, inputs)
    assert t.shape == (B,)
    got = full(t, c)
    # oracle on the whole batch (CPU: fine at this size)
    ops = E.make_arg_packer(tn)(tn.params, inputs)
    shapes = tuple(o.shape for o in ops)
    clist = E._contract_path(tn.einsum_str, shapes, optimize=path, memory_limit=None, use_blas=True)
    rt, rc, _ = cpu_ref.core_contract(list(ops), clist)
    ref = full(rt, rc)
    assert np.max(np.abs(got - ref)) <= 1e-3 * np.max(np.abs(ref))
    # batch independence: element j equals the contraction of input row j alone
    for j in (0, 517, B - 1):
        single = [np.ascontiguousarray(x[j:j + 1]) for x in inputs]
        tj, cj = fun(tn.params, single)
        assert abs(full(tj, cj)[0] - got[j]) <= 1e-3 * np.max(np.abs(ref))
    # the hyperedge is a batch index: no step materialises anything larger than B x bond
    plan = E._native_plan(clist, shapes, "float32")
    assert max(i["out_numel"] for i in plan.step_infos()) <= B * bond * phys
    infos = plan.step_infos()
    assert any(i["kernel"] == 2 and B in (i["m"], i["n"]) for i in infos)      # GEMM with M = batch
    # hyperedge steps: their own launches (batch label = B), or - the interior sites - folded into the GEMM's epilogue
    assert sum(i["batch"] == B or i["epilogue_sum"] == phys for i in infos) >= n_sites - 1
    assert sum(i["epilogue_sum"] == phys for i in infos) == n_sites - 2


def test_cfg4_cp_hyperedge_equals_tucker_with_delta_hub():
    r, n = 192, 160
    cp = nets.cp_network(TN, r, (n, n, n), dtype=np.float32, seed=5, scale=4.0)
    tk = nets.tucker_network(TN, (r, r, r), (n, n, n), dtype=np.float32, seed=5, scale=4.0, delta_hub=True)
    assert cp.einsum_str == "ac,ad,ae->cde"
    a = full(*cp.contract(split_format=True))
    b = full(*tk.contract(split_format=True))
    assert a.shape == (n, n, n)
    assert np.max(np.abs(a - b)) <= 1e-3 * np.max(np.abs(a))
    # spot-check entries against the definition sum_a A[a,c] B[a,d] C[a,e]
    A, Bm, C = [p.astype(np.float64) for p in cp.params]
    rng = np.random.default_rng(0)
    for c_, d_, e_ in rng.integers(0, n, size=(20, 3)):
        ref = np.sum(A[:, c_] * Bm[:, d_] * C[:, e_])
        assert abs(a[c_, d_, e_] - ref) <= 1e-3 * np.max(np.abs(a))


def test_cfg4_tucker_dense_hub_vs_numpy():
    tk = nets.tucker_network(TN, (96, 80, 64), (128, 96, 112), dtype=np.float32, seed=6, scale=8.0)
    got = full(*tk.contract(split_format=True))
    hub, m0, m1, m2 = [p.astype(np.float64) for p in tk.params]
    ref = np.einsum("abc,ae,bf,cg->efg", hub, m0, m1, m2, optimize=True)
    assert np.max(np.abs(got - ref)) <= 1e-3 * np.max(np.abs(ref))


@pytest.mark.parametrize("rows,cols,bond,dtype,tol", [(4, 4, 3, np.float64, 1e-9), (5, 5, 4, np.float32, 1e-3)])
def test_cfg5_peps_row_sweep_vs_oracle(rows, cols, bond, dtype, tol):
    tn = nets.peps_closed(TN, rows, cols, bond, dtype=dtype, seed=6)
    path = ssa_to_linear(nets.peps_row_path(rows, cols), 2 * rows * cols)
    t, c = tn.contract(optimize=path, split_format=True)
    rt, rc = cpu_ref.contract(tn.einsum_str, *tn.params, path=path, split_format=True)
    assert abs(float(t)) == 1.0 and float(t) == float(rt)
    assert abs(float(c) - float(rc)) <= tol * max(1.0, abs(float(rc)))


def test_cfg5_peps_sliced_equals_unsliced():
    """Index slicing (the multi-GPU decomposition) over three bulk bonds, world = 1."""
    rows = cols = 6
    tn = nets.peps_closed(TN, rows, cols, 4, dtype=np.float32, seed=6)
    path = ssa_to_linear(nets.peps_row_path(rows, cols), 2 * rows * cols)
    t, c = tn.contract(optimize=path, split_format=True)
    lhs, _ = tn.einsum_str.split("->")
    terms = lhs.split(",")
    # three vertical bonds in the middle of the grid (shared by site (2,c) and site (3,c))
    labels = tuple(next(iter(set(terms[2 * cols + k]) & set(terms[3 * cols + k]))) for k in (1, 2, 3))
    ts, cs = dist.contract_sliced(tn.einsum_str, list(tn.params), labels, optimize="greedy", rank=0, world=1)
    assert abs(full(ts, cs) - full(t, c)) <= 2e-3 * abs(full(t, c))


def test_torch_device_tensors_zero_copy():
    import torch

    g_ops = [torch.randn(64, 96, device="cuda"), torch.randn(96, 80, device="cuda")]
    t, c = contract("ab,bc->ac", *g_ops, split_format=True)
    assert t.is_cuda and t.dtype == torch.float32 and c.is_cuda
    ref = (g_ops[0].double() @ g_ops[1].double())
    got = t.double() * torch.exp(c.double())
    assert float((got - ref).abs().max() / ref.abs().max()) < 1e-5
    out = contract("ab,bc->ac", *g_ops)
    assert float((out.double() - ref).abs().max() / ref.abs().max()) < 1e-5
    # split_format=False on device tensors: de-stabilised in the SAME pass that divides by the last rescale factor
    # (ctn_exec_finish) - bit for bit `T_hat * exp(c)` with both roundings, dtype and device of the reference's torch backend
    assert out.is_cuda and out.dtype == torch.float32
    assert torch.equal(out, t * torch.exp(c.cpu()).to("cuda"))
    # ... an odd shape (scalar tail of the vector loop), fp64, and the overflow the reference has by design (README.md:73-74)
    a64 = torch.randn(33, 7, device="cuda", dtype=torch.float64)
    b64 = torch.randn(7, 13, device="cuda", dtype=torch.float64)
    t64, c64 = contract("ab,bc->ac", a64, b64, split_format=True)
    o64 = contract("ab,bc->ac", a64, b64)
    assert o64.dtype == torch.float64 and torch.equal(o64, t64 * torch.exp(c64.cpu()).to("cuda"))
    c_op = torch.randn(80, 24, device="cuda")
    big = contract("ab,bc,cd->ad", g_ops[0] * 1e15, g_ops[1] * 1e15, c_op * 1e15)    # every step in range, exp(register) not
    assert bool(torch.isinf(big).all())
    # the split format right after a plain call on the same cached executor: normalised again (mean |T_hat| = 1)
    t2, c2 = contract("ab,bc->ac", *g_ops, split_format=True)
    assert torch.equal(t2, t) and float(c2) == float(c)


def test_cfg5_device_resident_slicing_zero_copy():
    """Slices as replicas of one plan (pointer offsets into the uploaded tensors): same answer as
    the unsliced contraction; two 'ranks' emulated in-process partition the slices exactly."""
    rows = cols = 5
    tn = nets.peps_closed(TN, rows, cols, 4, dtype=np.float32, seed=6)
    path = ssa_to_linear(nets.peps_row_path(rows, cols), 2 * rows * cols)
    t, c = tn.contract(optimize=path, split_format=True)
    terms = tn.einsum_str.split("->")[0].split(",")
    labels = tuple(next(iter(set(terms[2 * cols + k]) & set(terms[3 * cols + k]))) for k in (1, 2, 3))
    whole = dist.SlicedContraction(tn.einsum_str, list(tn.params), labels, optimize=path, rank=0, world=1)
    assert whole.n_total == 64 and len(whole.my_slices) == 64
    ts, cs = whole.run()
    assert abs(full(ts, cs) - full(t, c)) <= 2e-3 * abs(full(t, c))
    parts = []
    for rank in range(2):
        sc = dist.SlicedContraction(tn.einsum_str, list(tn.params), labels, optimize=path, rank=rank, world=2)
        assert len(sc.my_slices) == 32
        parts.append(sc.local_result())
    tj, cj = dist.combine_split(parts)
    assert abs(full(tj, cj) - full(t, c)) <= 2e-3 * abs(full(t, c))
    # a tiny workspace budget forces several groups (incl. a short, padded last one)
    small = dist.SlicedContraction(tn.einsum_str, list(tn.params), labels, optimize=path, workspace_budget=1)
    assert small.R == 1 and len(small._chunks) == 64
    tk, ck = small.run()
    assert abs(full(tk, ck) - full(t, c)) <= 2e-3 * abs(full(t, c))


def test_cfg5_peps_on_the_librarys_own_path_and_co_optimised_slices():
    """6 x 6 PEPS, bond 4: `optimize="auto"` (noisy greedy + subtree reconfiguration) and slices chosen together
    with their path (`dist.choose_slices_with_path`, run device-resident as replicas) give the row sweep's value."""
    rows = cols = 6
    tn = nets.peps_closed(TN, rows, cols, 4, dtype=np.float32, seed=6)
    ops = list(tn.params)
    row = ssa_to_linear(nets.peps_row_path(rows, cols), 2 * rows * cols)
    t, c = tn.contract(optimize=row, split_format=True)
    ta, ca = tn.contract(optimize="auto", split_format=True)
    assert float(ta) == float(t) and abs(float(ca) - float(c)) <= 2e-5 * abs(float(c))
    labels, path, rep = dist.choose_slices_with_path(tn.einsum_str, [o.shape for o in ops], min_slices=16, trials=2)
    assert rep["slices"] >= 16
    sc = dist.SlicedContraction(tn.einsum_str, ops, labels, optimize=path, rank=0, world=1)
    ts, cs = sc.run()
    assert abs(full(ts, cs) - full(t, c)) <= 2e-3 * abs(full(t, c))
