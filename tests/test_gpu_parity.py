"""HIP engine vs the reference's golden vectors and vs the CPU oracle (needs a GPU).

Tolerances (BASELINE.json north_star): 1e-6 relative in fp64, 1e-3 in fp32;
the log-scale register is bit-exact where the reference's abs-sums are exact
(README examples: two-element / all-ones tensors).
"""
import numpy as np
import pytest

from tests.helpers import golden_names, load_golden
from contractn_amd import contract, engine
from contractn_amd import einsum as E

pytestmark = pytest.mark.gpu

RTOL = {np.dtype(np.float64): 1e-6, np.dtype(np.float32): 1e-3}
# measured headroom is far larger; these are the contract, asserted tighter below where cheap
TIGHT = {np.dtype(np.float64): 1e-11, np.dtype(np.float32): 2e-5}


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    denom = max(np.max(np.abs(b)), 1e-300)
    return float(np.max(np.abs(a - b)) / denom)


def test_device_present():
    assert engine.device_count() >= 1, "no HIP device: the engine has no CPU fallback"


@pytest.mark.parametrize("name", golden_names())
def test_golden_split(name):
    g = load_golden(name)
    ops = g["operands"]
    t_hat, log_scale = contract(g["einsum_str"], *ops, optimize=g["path"], split_format=True)
    dt = np.result_type(*[o.dtype for o in ops])
    dt = np.dtype(np.float32) if dt in (np.float32, np.complex64) else np.dtype(np.float64)   # (complex: its components' type)
    assert t_hat.shape == g["t_hat"].shape
    assert t_hat.dtype == g["t_hat"].dtype
    assert isinstance(log_scale, np.ndarray) and log_scale.shape == () and log_scale.dtype == np.float64
    err = rel_err(t_hat, g["t_hat"])
    assert err <= RTOL[dt], (name, err)
    assert err <= TIGHT[dt], (name, err)
    ref_ls = float(g["log_scale"])
    tol = 1e-10 * max(1.0, abs(ref_ls)) if dt == np.float64 else 2e-5 * max(1.0, abs(ref_ls))
    assert abs(float(log_scale) - ref_ls) <= tol, (name, float(log_scale), ref_ls)


@pytest.mark.parametrize("name", ["readme_copy101", "readme_chain1000", "edge_zero", "mps_open_ones_f64"])
def test_log_scale_bit_exact(name):
    """Exact abs-sums => the register must match the reference bit for bit."""
    g = load_golden(name)
    _, log_scale = contract(g["einsum_str"], *g["operands"], optimize=g["path"], split_format=True)
    assert float(log_scale).hex() == g["log_scale_hex"]


@pytest.mark.parametrize("name", ["readme_copy101", "readme_chain1000", "cp_r5_f32", "mps_overlap_6x8x3_f32"])
def test_golden_plain(name):
    """split_format=False: destabilised product, float64 for float32 inputs, inf on overflow."""
    g = load_golden(name)
    with np.errstate(over="ignore"):
        out = contract(g["einsum_str"], *g["operands"], optimize=g["path"])
    assert out.dtype == g["plain"].dtype
    if np.all(np.isinf(g["plain"])):
        assert np.all(np.isinf(out))
    else:
        assert rel_err(out, g["plain"]) <= 1e-5


def test_readme_known_answers():
    g = load_golden("readme_copy101")
    out = contract(g["einsum_str"], *g["operands"])
    np.testing.assert_allclose(out, [1.0, 0.99 ** 100], rtol=1e-12)
    g = load_golden("readme_chain1000")
    t, c = contract(g["einsum_str"], *g["operands"], optimize=g["path"], split_format=True)
    np.testing.assert_array_equal(t, [1.0, 1.0, 1.0])
    assert float(c).hex() == "0x1.12a72fbccf574p+10"


def test_auto_path_matches_explicit():
    g = load_golden("peps3x3_D2_f64")
    t1, c1 = contract(g["einsum_str"], *g["operands"], split_format=True)
    full1 = float(t1) * np.exp(float(c1))
    full0 = float(g["t_hat"]) * np.exp(float(g["log_scale"]))
    assert abs(full1 - full0) <= 1e-10 * abs(full0)


def test_unknown_kwarg_raises():
    with pytest.raises(TypeError):
        contract("a,a->a", np.ones(2), np.ones(2), bogus=1)


def test_replicas_batched_execution():
    """R independent contractions in one launch sequence give the same results as R single runs."""
    g = load_golden("mps_overlap_5x64x4_f32")
    shapes = tuple(o.shape for o in g["operands"])
    clist = E._contract_path(g["einsum_str"], shapes, optimize=g["path"], memory_limit=None, use_blas=True)
    plan = E._native_plan(clist, shapes, "float32")
    R = 3
    rng = np.random.default_rng(0)
    sets = [[(o * rng.uniform(0.5, 1.5)).astype(np.float32) for o in g["operands"]] for _ in range(R)]
    ex = engine.Executor(plan, replicas=R)
    outs, logs, resc = ex.run_host(sets)
    ex1 = engine.Executor(plan, replicas=1)
    for r in range(R):
        o1, l1, r1 = ex1.run_host([sets[r]])
        np.testing.assert_array_equal(outs[r], o1[0])
        np.testing.assert_array_equal(resc[r], r1[0])
        assert logs[r] == l1[0]
    assert any(i["kernel"] == 2 for i in plan.step_infos()), "expected MFMA steps in this plan"


def test_f64_plan_uses_f64_mfma_kernel():
    g = load_golden("mps_overlap_4x48x4_f64")
    shapes = tuple(o.shape for o in g["operands"])
    clist = E._contract_path(g["einsum_str"], shapes, optimize=g["path"], memory_limit=None, use_blas=True)
    plan = E._native_plan(clist, shapes, "float64")
    assert any(i["kernel"] == 3 for i in plan.step_infos())


@pytest.mark.parametrize("dtype,tol", [(np.float32, 2e-5), (np.float64, 1e-12)])
@pytest.mark.parametrize("einstr,shapes", [
    ("ab,bc->ac", [(70, 33), (33, 129)]),            # ragged: no dimension a multiple of the tile
    ("ba,bc->ac", [(40, 200), (40, 36)]),            # A column-major
    ("ab,cb->ac", [(130, 64), (96, 64)]),            # both k-contiguous
    ("abc,cbd->ad", [(48, 6, 10), (10, 6, 52)]),     # two contracted labels, different orders
    ("xab,xbc->xac", [(3, 40, 32), (3, 32, 48)]),    # batch (hyperedge) label on an MFMA step
    ("ab,bc->ca", [(64, 64), (64, 128)]),            # transposed output (operand swap)
])
def test_single_step_shapes_vs_numpy(dtype, tol, einstr, shapes):
    rng = np.random.default_rng(1)
    ops = [rng.standard_normal(s).astype(dtype) for s in shapes]
    t_hat, c = contract(einstr, *ops, split_format=True)
    ref = np.einsum(einstr, *[o.astype(np.float64) for o in ops])
    got = t_hat.astype(np.float64) * np.exp(float(c))
    assert got.shape == ref.shape
    assert np.max(np.abs(got - ref)) <= tol * np.max(np.abs(ref)) * 10
    assert abs(np.mean(np.abs(t_hat)) - 1.0) < 1e-5  # stabilised: mean |T| == 1


def test_dtype_promotion_follows_reference_table():
    """ints and mixed precisions compute in float64; float32 stays float32."""
    a, b = np.arange(6).reshape(2, 3), np.arange(12).reshape(3, 4)
    t, c = contract("ab,bc->ac", a, b, split_format=True)
    assert t.dtype == np.float64
    np.testing.assert_allclose(t * np.exp(c), a @ b, rtol=1e-13)
    t, c = contract("ab,bc->ac", a.astype(np.float32), b.astype(np.float64), split_format=True)
    assert t.dtype == np.float64
    t, c = contract("ab,bc->ac", a.astype(np.float32), b.astype(np.float32), split_format=True)
    assert t.dtype == np.float32 and c.dtype == np.float64
    out = contract("ab,bc->ac", a.astype(np.float32), b.astype(np.float32))
    assert out.dtype == np.float64  # float32 tensor times the 0-d float64 register (reference behaviour)
    t, _ = contract("ab,bc->ac", a, b, split_format=True, dtype=np.float32)
    assert t.dtype == np.float32


def test_non_contiguous_and_list_paths():
    rng = np.random.default_rng(3)
    a = rng.standard_normal((5, 7)).T            # Fortran-ordered view
    b = rng.standard_normal((10, 5))[::2]        # strided rows
    c = rng.standard_normal((5, 4))
    ref = np.einsum("ab,cb,cd->ad", a, b, c)
    for path in ([(0, 1), (0, 1)], ((1, 2), (0, 1)), "greedy", "optimal", True):
        out = contract("ab,cb,cd->ad", a, b, c, optimize=path)
        np.testing.assert_allclose(out, ref, rtol=1e-12)
    out = contract("ab,cb,cd->ad", a, b, c, memory_limit=None, use_blas=False, order="K", casting="safe")
    np.testing.assert_allclose(out, ref, rtol=1e-12)
    with pytest.raises(NotImplementedError):
        contract("ab,bc->ac", a, a.T, backend="jax")


def test_make_contract_fun_accepts_new_params_each_call():
    """reference ctn.py:349-387 / SURVEY App. C-13: the compiled closure takes fresh tensors."""
    from contractn_amd import TN

    rng = np.random.default_rng(4)
    tn = TN()
    x = tn.add_dense_node(rng.standard_normal((6, 5)))
    y = tn.add_dense_node(rng.standard_normal((5, 3)))
    tn.connect_nodes(x, y, 1, 0)
    fun = tn.make_contract_fun()
    for _ in range(3):
        p = (rng.standard_normal((6, 5)), rng.standard_normal((5, 3)))
        np.testing.assert_allclose(fun(p, ()), p[0] @ p[1], rtol=1e-12)
    x.tensor = np.ones((6, 5))
    np.testing.assert_allclose(tn.contract(), np.ones((6, 5)) @ tn.params[1], rtol=1e-12)


def test_executor_cache_is_bounded_and_clearable():
    import contractn_amd
    from contractn_amd import einsum as EE

    contractn_amd.clear_caches()
    rng = np.random.default_rng(0)
    for n in range(3, 3 + EE.MAX_CACHED_EXECUTORS + 5):
        a, b = rng.standard_normal((n, 4)), rng.standard_normal((4, n))
        np.testing.assert_allclose(contract("ab,bc->ac", a, b), a @ b, rtol=1e-12)
    assert len(EE._EXECUTOR_LRU) == EE.MAX_CACHED_EXECUTORS
    contractn_amd.clear_caches()
    assert len(EE._EXECUTOR_LRU) == 0
    a = rng.standard_normal((5, 4))
    np.testing.assert_allclose(contract("ab,cb->ac", a, a), a @ a.T, rtol=1e-12)


def test_clone_nodes_contract_with_shared_weights():
    """Weight sharing (broken in the reference, works here): a clone node is the
    same tensor appearing twice in the einsum."""
    from contractn_amd import TN

    rng = np.random.default_rng(8)
    w = rng.standard_normal((4, 6))
    tn = TN()
    base = tn.add_dense_node(w)
    clone = tn.add_duplicate_node(base)
    tn.connect_nodes(base, clone, 1, 1)
    assert tn.einsum_str == "ab,cb->ac" and len(tn.params) == 1
    np.testing.assert_allclose(tn.contract(), w @ w.T, rtol=1e-12)
    t, c = tn.make_contract_fun(split_format=True)((2.0 * w,), ())
    np.testing.assert_allclose(t * np.exp(c), 4.0 * (w @ w.T), rtol=1e-12)


def test_tn_level_batched_contract_fun():
    from contractn_amd import TN
    from contractn_amd.paths import ssa_to_linear
    from tests import networks as nets

    tn, ssa = nets.mps_overlap(TN, 5, 16, 3, dtype=np.float64, seed=1)
    path = ssa_to_linear(ssa, 10)
    single = tn.make_contract_fun(optimize=path, split_format=True)
    batched = tn.make_batched_contract_fun(4, optimize=path)
    rng = np.random.default_rng(2)
    plist = [tuple(p * rng.uniform(0.5, 2.0) for p in tn.params) for _ in range(4)]
    t, c = batched(plist)
    assert t.shape == (4,) and c.shape == (4,)
    for r in range(4):
        t1, c1 = single(plist[r], ())
        assert float(t[r]) == float(t1) and float(c[r]) == float(c1)


# --- the large-tile LDS-DMA MFMA kernel (k_mfma_f32_g): steps with both operands unit-stride along
# their free index, M % 256 == 0, N % 128 == 0, K % 16 == 0, K >= 32 --------------------------------
def _g_eligible(info):
    return info["kernel"] == 2 and info["tile_m"] == 256   # the planner's decision (plan.cpp)


@pytest.fixture
def force_large_tiles(monkeypatch):
    """By default the launcher keeps small launches (< 2 big tiles per CU) on 128-row tiles; these tests
    are about the large-tile kernel, so executors created inside them take it whenever a step is eligible."""
    monkeypatch.setenv("CTN_MFMA_G", "2")
    E.clear_caches()
    yield
    E.clear_caches()


@pytest.mark.parametrize("einstr,shapes,path", [
    ("km,kn->mn", [(32, 256), (32, 128)], None),                    # one tile, two k-tiles (ring minimum)
    ("km,kn->mn", [(48, 512), (48, 384)], None),                    # 2 x 3 tiles, three k-tiles
    ("km,kn->mn", [(272, 256), (272, 256)], None),                  # 17 k-tiles: ring wraps many times
    ("xkm,xkn->xmn", [(3, 64, 256), (3, 64, 128)], None),           # batch (hyperedge) label
    ("kam,kbn->ambn", [(64, 2, 128), (64, 2, 64)], None),           # composite free indices, strided C rows
    ("km,kn,nj->mj", [(64, 256), (64, 256), (256, 256)], ((0, 1), (0, 1))),  # an operand = rescaled intermediate
    ("km,kn->mn", [(48, 400), (48, 200)], None),                    # ragged M and N: masked edge tiles
    ("km,kn->mn", [(32, 896), (32, 196)], None),                    # 3.5 row tiles, ragged columns
    ("xkm,xkn->xmn", [(2, 64, 508), (2, 64, 128)], None),           # batch + 4 masked rows
    ("km,kn->mn", [(33, 256), (33, 128)], None),                    # ragged K: 1 valid row in the last k-tile
    ("km,kn->mn", [(100, 256), (100, 256)], None),                  # K = 6 k-tiles + 4
    ("kam,kbn->ambn", [(47, 4, 100), (47, 2, 100)], None),          # everything ragged at once
    ("mk,kn->mn", [(256, 32), (32, 128)], None),                    # row-major A (k-contiguous): requests along k
    ("km,nk->mn", [(48, 512), (384, 48)], None),                    # k-contiguous B
    ("mk,nk->mn", [(400, 64), (200, 64)], None),                    # both, ragged M and N
    ("amk,kbn->ambn", [(4, 100, 32), (32, 2, 100)], None),          # composite rows over a k-contiguous A
    ("xmk,xkn->xmn", [(2, 256, 272), (2, 272, 128)], None),         # batch + 17 k-tiles
    ("mk,kn->mn", [(256, 40), (40, 128)], None),                    # row-major A with a ragged K (C++ inner loop)
    ("mk,nk->mn", [(400, 100), (200, 100)], None),                  # both k-contiguous, K = 6 k-tiles + 4
    ("km,nk->mn", [(36, 512), (128, 36)], None),                    # k-contiguous B, 4 valid rows in the last k-tile
])
def test_large_tile_dma_kernel_vs_numpy(einstr, shapes, path, force_large_tiles):
    rng = np.random.default_rng(7)
    ops = [(rng.standard_normal(s) * rng.uniform(0.5, 3.0)).astype(np.float32) for s in shapes]
    kw = {"optimize": path} if path is not None else {}
    clist = E._contract_path(einstr, tuple(shapes), optimize=path if path is not None else "auto",
                             memory_limit=None, use_blas=True)
    infos = E._native_plan(clist, tuple(shapes), "float32").step_infos()
    assert _g_eligible(infos[-1]), infos[-1]       # keeps this test on the kernel it is about
    t_hat, c = contract(einstr, *ops, split_format=True, **kw)
    ref = np.einsum(einstr, *[o.astype(np.float64) for o in ops])
    got = t_hat.astype(np.float64) * np.exp(float(c))
    assert got.shape == ref.shape
    assert np.max(np.abs(got - ref)) <= 2e-5 * np.max(np.abs(ref)) * 10
    assert abs(np.mean(np.abs(t_hat)) - 1.0) < 1e-5


def test_large_tile_dma_kernel_replicas_and_exact_sums(force_large_tiles):
    # small integers: every product and partial sum is exact in fp32, so the result must be bit-identical
    # to NumPy and identical across replicas holding the same data
    rng = np.random.default_rng(11)
    A = rng.integers(-3, 4, size=(96, 256)).astype(np.float32)
    B = rng.integers(-3, 4, size=(96, 256)).astype(np.float32)
    bc = E.BatchedContraction("km,kn->mn", [A.shape, B.shape], np.float32, optimize=((0, 1),), replicas=3)
    assert _g_eligible(bc.plan.step_infos()[0])
    t, c = bc.run_host([[A, B], [2 * A, B], [A, B]])
    ref = A.T.astype(np.float64) @ B.astype(np.float64)
    for r, f in enumerate((1.0, 2.0, 1.0)):
        got = t[r].astype(np.float64) * np.exp(float(c[r]))
        assert np.max(np.abs(got - f * ref)) <= 1e-5 * np.max(np.abs(ref))
    assert np.array_equal(t[0], t[2]) and float(c[0]) == float(c[2])
    norm = np.abs(ref).sum()
    assert np.allclose(t[0].astype(np.float64), ref / (norm / ref.size), rtol=3e-7, atol=0)


# --- the fp64 sibling (k_mfma_f64_g): 128 x 128 tiles, LDS-DMA ring ------------------------------------
@pytest.mark.parametrize("einstr,shapes,path", [
    ("km,kn->mn", [(16, 128), (16, 128)], None),                    # one tile, two k-tiles (ring minimum)
    ("km,kn->mn", [(40, 256), (40, 256)], None),                    # 2 x 2 tiles, five k-tiles
    ("km,kn->mn", [(19, 130), (19, 250)], None),                    # ragged M, N and K (3 valid rows in the last k-tile)
    ("km,kn->mn", [(17, 128), (17, 128)], None),                    # one valid row in the last k-tile
    ("xkm,xkn->xmn", [(2, 32, 128), (2, 32, 128)], None),           # batch (hyperedge) label
    ("kam,kbn->ambn", [(24, 2, 64), (24, 2, 64)], None),            # composite free indices
    ("km,kn,nj->mj", [(64, 128), (64, 128), (128, 128)], ((0, 1), (0, 1))),  # an operand = rescaled intermediate
    ("mk,kn->mn", [(128, 16), (16, 128)], None),                    # row-major A (k-contiguous): requests along k
    ("km,nk->mn", [(40, 256), (256, 40)], None),                    # k-contiguous B
    ("mk,nk->mn", [(130, 20), (250, 20)], None),                    # both, ragged M, N and K (4 valid rows in the last k-tile)
    ("mk,kn->mn", [(128, 18), (18, 128)], None),                    # K tail of 2
    ("xmk,xkn->xmn", [(2, 128, 72), (2, 72, 128)], None),           # batch + 9 k-tiles
])
def test_large_tile_dma_kernel_f64_vs_numpy(einstr, shapes, path, force_large_tiles):
    rng = np.random.default_rng(17)
    ops = [rng.standard_normal(s) * rng.uniform(0.5, 3.0) for s in shapes]
    kw = {"optimize": path} if path is not None else {}
    clist = E._contract_path(einstr, tuple(shapes), optimize=path if path is not None else "auto",
                             memory_limit=None, use_blas=True)
    info = E._native_plan(clist, tuple(shapes), "float64").step_infos()[-1]
    assert info["kernel"] == 3 and info["tile_n"] == 128, info     # keeps this test on the kernel it is about
    t_hat, c = contract(einstr, *ops, split_format=True, **kw)
    ref = np.einsum(einstr, *ops)
    got = t_hat * np.exp(float(c))
    assert got.shape == ref.shape
    assert np.max(np.abs(got - ref)) <= 1e-12 * np.max(np.abs(ref)) * 10
    assert abs(np.mean(np.abs(t_hat)) - 1.0) < 1e-13


def test_launcher_retiles_by_replica_count():
    """The planner's tile is a default: the launcher takes 256 x 256 tiles for long-K full steps once there is
    a tile per CU, and the one-launch latency form's 16 x 16 tiles for a single small network (ctn_exec_step_tile)."""
    rng = np.random.default_rng(23)
    A = (rng.standard_normal((1024, 256)) / 8).astype(np.float32)
    B = (rng.standard_normal((1024, 256)) / 8).astype(np.float32)
    ref = A.T.astype(np.float64) @ B.astype(np.float64)
    for replicas, tile in ((1, (16, 16)), (256, (256, 256))):
        bc = E.BatchedContraction("km,kn->mn", [A.shape, B.shape], np.float32, optimize=((0, 1),), replicas=replicas)
        t, c = bc.run_host([[A, B]] * replicas)
        assert bc.executor.step_tiles() == [tile]
        for r in (0, replicas - 1):
            got = t[r].astype(np.float64) * np.exp(float(c[r]))
            assert np.max(np.abs(got - ref)) <= 2e-5 * np.max(np.abs(ref)) * 10
        assert np.array_equal(t[0], t[replicas - 1])
        bc.executor.close()


@pytest.mark.parametrize("dtype,tol", [(np.float32, 2e-5), (np.float64, 1e-12)])
@pytest.mark.parametrize("einstr,shapes", [
    ("km,kn->mn", [(300, 70), (300, 100)]),             # ragged 64-tiles, 10 k-tiles of 32 minus a bit
    ("mk,kn->mn", [(256, 1024), (1024, 256)]),          # the MPS environment step, row-major A
    ("mk,nk->mn", [(130, 200), (90, 200)]),             # both k-contiguous
    ("xkm,xkn->xmn", [(2, 160, 64), (2, 160, 96)]),     # batch label
    ("km,kn->mn", [(512, 128), (512, 128)]),            # 8 slabs (the reduce pass's 8 x 2 form)
    ("km,kn->mn", [(2304, 64), (2304, 64)]),            # 36 slabs: more than the unrolled forms take
    ("km,kn->mn", [(256, 65), (256, 33)]),              # odd element count: scalar reduce path
    ("km,kn->mn", [(19200, 64), (19200, 64)]),          # 300 slabs: folded 16 to 1 twice (300 -> 19 -> 2)
    ("km,kn->mn", [(2304, 65), (2304, 33)]),            # 36 slabs of an odd element count: scalar fold
])
def test_latency_mode_split_k(dtype, tol, einstr, shapes, monkeypatch):
    """One small network: 64 x 64 tiles with K split over workgroups and a fixed-order slab reduction,
    in fp32 and in fp64 (ctn_exec_step_tile reports the 64 x 64 kernel).  (fp32 steps with at most 64 small tiles
    take the one-launch form by default - test_latency_mode_one_launch; here it is switched off.)"""
    monkeypatch.setenv("CTN_LAT", "0")
    E.clear_caches()
    rng = np.random.default_rng(5)
    ops = [rng.standard_normal(s).astype(dtype) for s in shapes]
    bc = E.BatchedContraction(einstr, shapes, dtype, optimize=((0, 1),), replicas=1)
    t, c = bc.run_host([ops])
    assert bc.executor.step_tiles() == [(64, 64)]
    bc.executor.close()
    ref = np.einsum(einstr, *[o.astype(np.float64) for o in ops])
    got = t[0].astype(np.float64) * np.exp(float(c[0]))
    assert np.max(np.abs(got - ref)) <= tol * np.max(np.abs(ref)) * 10
    t2, c2 = contract(einstr, *ops, split_format=True)      # bit-reproducible: same reduction order every time
    assert np.array_equal(t2, t[0]) and float(c2) == float(c[0])
    E.clear_caches()


@pytest.mark.parametrize("einstr,shapes,replicas,tile", [
    ("mk,kn->mn", [(256, 1024), (1024, 256)], 1, (16, 16)),    # the MPS environment step alone: 256 tiles of 16 x 16 fill the chip
    ("mk,kn->mn", [(256, 1024), (1024, 256)], 4, (32, 32)),    # ... four networks in flight: 4 x 64 tiles of 32 x 32
    ("km,kn->mn", [(256, 256), (256, 1024)], 1, (32, 32)),     # the other zipper step: 256 tiles of 32 x 32
    ("km,kn->mn", [(256, 256), (256, 1024)], 4, (64, 64)),     # ... 4 x 64 tiles of 64 x 64, two K halves per block
    ("km,kn->mn", [(300, 70), (300, 100)], 1, (16, 16)),       # ragged M, N and K (k chunks of 40: leftover groups + tail)
    ("mk,nk->mn", [(130, 200), (90, 200)], 2, (16, 16)),       # both k-contiguous, ragged tiles
    ("xkm,xkn->xmn", [(2, 160, 64), (2, 160, 96)], 1, (16, 16)),  # batch label: 2 x 4 x 6 tiles
    ("kam,kbn->ambn", [(136, 4, 24), (136, 2, 40)], 1, (16, 16)),  # composite free indices, strided C
    ("km,kn->mn", [(4096, 64), (4096, 64)], 1, (16, 16)),      # K = 4096: both tables fill their LDS arrays
    ("km,kn->mn", [(130, 256), (130, 256)], 4, (32, 32)),      # 4 x 64 tiles of 32: odd k chunk (17 -> 18)
    ("km,kn->mn", [(130, 512), (130, 512)], 4, (64, 64)),      # 4 x 64 tiles of 64 (one per CU): k chunk 65 -> 66
])
def test_latency_mode_one_launch(einstr, shapes, replicas, tile):
    """A few networks in flight, fp32: the step runs as ONE launch of k_mfma_f32_lat - K split over the eight waves
    of a workgroup, partials added in wave order in LDS, the tile (16 / 32 / 64) the largest that still gives every
    CU a workgroup - instead of split-K slabs plus a reduce launch."""
    rng = np.random.default_rng(6)
    sets = [[rng.standard_normal(s).astype(np.float32) for s in shapes] for _ in range(replicas)]
    bc = E.BatchedContraction(einstr, shapes, np.float32, optimize=((0, 1),), replicas=replicas)
    t, c = bc.run_host(sets)
    assert bc.executor.step_tiles() == [tile]
    for r in range(replicas):
        ref = np.einsum(einstr, *[o.astype(np.float64) for o in sets[r]])
        got = t[r].astype(np.float64) * np.exp(float(c[r]))
        assert np.max(np.abs(got - ref)) <= 2e-4 * np.max(np.abs(ref))
        assert abs(np.mean(np.abs(t[r])) - 1.0) < 1e-5
    t2, c2 = bc.run_host(sets)                               # fixed reduction order: the same bits every time
    assert np.array_equal(t2, t) and np.array_equal(c2, c)
    bc.executor.close()
    # small exact integers: every product and partial sum is exact, so the result is NumPy's
    iops = [rng.integers(-3, 4, size=s).astype(np.float32) for s in shapes]
    ti, ci = contract(einstr, *iops, split_format=True)
    refi = np.einsum(einstr, *[o.astype(np.float64) for o in iops])
    np.testing.assert_allclose(ti.astype(np.float64) * np.exp(float(ci)), refi, rtol=1e-6, atol=1e-6 * np.max(np.abs(refi)))


@pytest.mark.parametrize("einstr,shapes,replicas", [
    ("mk,kn->mn", [(256, 1024), (1024, 256)], 1),      # the MPS environment step in double precision
    ("km,kn->mn", [(300, 70), (300, 100)], 1),         # ragged everything
    ("xkm,xkn->xmn", [(2, 160, 64), (2, 160, 96)], 2),  # batch label, two networks in flight
    ("mk,nk->mn", [(130, 200), (90, 200)], 1),         # both k-contiguous
])
def test_latency_mode_one_launch_fp64(einstr, shapes, replicas):
    """The same one-launch form in double precision: 16 x 16 tiles on v_mfma_f64_16x16x4_f64 (its accumulator rows
    are laid out differently from the fp32 one: a transposed store would pass no test with a symmetric product)."""
    rng = np.random.default_rng(8)
    sets = [[rng.standard_normal(s) for s in shapes] for _ in range(replicas)]
    bc = E.BatchedContraction(einstr, shapes, np.float64, optimize=((0, 1),), replicas=replicas)
    t, c = bc.run_host(sets)
    assert bc.executor.step_tiles() == [(16, 16)]
    for r in range(replicas):
        ref = np.einsum(einstr, *sets[r])
        got = t[r] * np.exp(float(c[r]))
        assert np.max(np.abs(got - ref)) <= 1e-12 * np.max(np.abs(ref))
    t2, c2 = bc.run_host(sets)
    assert np.array_equal(t2, t) and np.array_equal(c2, c)
    bc.executor.close()


def test_latency_mode_feeds_rescaled_intermediates(monkeypatch):
    """A chain of latency-form steps (forced also for its short-K steps): every step consumes its predecessor's
    un-normalised output and abs-sum partials (one per tile) exactly like the throughput kernels' outputs."""
    monkeypatch.setenv("CTN_LAT", "1")
    E.clear_caches()
    g = load_golden("mps_overlap_5x64x4_f32")
    plan, ex = _plan_and_executor(g)
    outs, _log, resc = ex.run_host([g["operands"]])
    tiles = ex.step_tiles()
    assert sum(t in ((16, 16), (32, 32)) for t in tiles) >= 6, tiles
    c = E.accumulate_log_scale(resc[0], np.dtype(np.float32))
    assert float(outs[0]) == float(g["t_hat"]) and abs(float(c) - float(g["log_scale"])) <= 2e-5 * abs(float(g["log_scale"]))
    E.clear_caches()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape", [(), (7,), (33, 65), (4, 3, 5, 2), (1024, 1024)])
def test_stabilize_function(shape, dtype):
    """The stand-alone ``stabilize`` (reference einsum.py:89-107): rescaled tensor, register update,
    exact-sum cases bit for bit, and the "norm below 1e-7 leaves both unchanged" branch."""
    from oracle import cpu_ref

    rng = np.random.default_rng(len(shape) + 17)
    t = np.asarray(rng.standard_normal(shape) * 37.5).astype(dtype)
    c0 = np.asarray(2.5)
    got_t, got_c = E.stabilize(t, c0)
    ref_t, ref_c = cpu_ref.stabilize(t, c0)
    assert got_t.shape == t.shape and got_t.dtype == t.dtype
    assert rel_err(got_t, ref_t) <= TIGHT[np.dtype(dtype)]
    assert abs(float(got_c) - float(ref_c)) <= (1e-12 if dtype == np.float64 else 2e-6) * abs(float(ref_c))

    ones = np.full(shape, 8.0, dtype=dtype)  # power-of-two values: every abs-sum is exact
    got_t, got_c = E.stabilize(ones, np.asarray(0.0))
    ref_t, ref_c = cpu_ref.stabilize(ones, np.asarray(0.0))
    np.testing.assert_array_equal(got_t, ref_t)
    assert float(got_c).hex() == float(ref_c).hex()

    tiny = np.full(shape, 1e-12, dtype=dtype) if np.prod(shape, dtype=np.int64) * 1e-12 < 1e-7 else np.zeros(shape, dtype)
    got_t, got_c = E.stabilize(tiny, np.asarray(1.25))
    np.testing.assert_array_equal(got_t, tiny)
    assert float(got_c) == 1.25


def test_graph_replay_matches_eager_and_follows_new_operands(monkeypatch):
    """From its third enqueue on an executor replays the launch sequence as one hipGraph: same bits as the
    eager enqueues, new operand values (same buffers or new ones) are picked up, timing mode stays eager."""
    from tests import networks as nets
    from contractn_amd import TN

    tn = nets.peps_closed(TN, 3, 3, 6, dtype=np.float32, seed=4)   # bond 6: too big for the one-launch chain walk
    ops = [np.asarray(p, dtype=np.float32) for p in tn.params]
    shapes = [o.shape for o in ops]
    bc = E.BatchedContraction(tn.einsum_str, shapes, np.float32, replicas=2)
    assert any(i["kernel"] == 2 for i in bc.plan.step_infos())   # MFMA steps: not a chain-walk plan
    sets = [ops, [2 * o if i == 0 else o for i, o in enumerate(ops)]]
    runs = [bc.run_host(sets) for _ in range(5)]           # eager, eager (capture), replay, replay, replay
    for t, c in runs[1:]:
        assert np.array_equal(t, runs[0][0]) and np.array_equal(c, runs[0][1])
    sets2 = [[3 * o if i == 1 else o for i, o in enumerate(ops)], ops]
    t2, c2 = bc.run_host(sets2)                            # replayed graph, different operand values
    assert np.array_equal(t2[1], runs[0][0][0]) and float(c2[1]) == float(runs[0][1][0])
    assert abs(float(c2[0]) - (float(runs[0][1][0]) + np.log(3.0))) < 1e-5
    bc.executor.set_timing(1)
    t3, c3 = bc.run_host(sets)                             # event-timed enqueue: eager path, same result
    assert np.array_equal(t3, runs[0][0]) and np.array_equal(c3, runs[0][1])
    assert bc.executor.step_ms().shape == (bc.plan.n_steps,)
    bc.executor.close()

    monkeypatch.setenv("CTN_GRAPH", "0")
    bc0 = E.BatchedContraction(tn.einsum_str, shapes, np.float32, replicas=2)
    t0, c0 = bc0.run_host(sets)
    assert np.array_equal(t0, runs[0][0]) and np.array_equal(c0, runs[0][1])
    bc0.executor.close()


@pytest.mark.parametrize("dtype,tol", [(np.float32, 2e-5), (np.float64, 1e-12)])
@pytest.mark.parametrize("einstr,shapes", [
    ("abc,abc->", [(64, 64, 64), (64, 64, 64)]),          # the inner product that closes a network: 1 output, K = 262,144
    ("ab,ba->", [(1000, 777), (777, 1000)]),              # K = 777,000 with one operand transposed (gathered k)
    ("ka,kb->ab", [(100000, 8), (100000, 8)]),            # 64 outputs, K = 100,000
    ("xk,xk->x", [(3, 40000), (3, 40000)]),               # batch label, 3 outputs
])
def test_huge_k_tiny_output_is_split_over_workgroups(dtype, tol, einstr, shapes):
    """At most 64 outputs and K >= 32768: K is split over workgroups (k_dot_split) and the slabs are added by the
    split-K reduce pass in a fixed order - same value as NumPy, same bits run to run, also inside a path."""
    rng = np.random.default_rng(13)
    ops = [rng.standard_normal(s).astype(dtype) for s in shapes]
    t, c = contract(einstr, *ops, split_format=True)
    ref = np.einsum(einstr, *[o.astype(np.float64) for o in ops])
    terms = np.einsum(einstr, *[np.abs(o).astype(np.float64) for o in ops])
    got = np.asarray(t, dtype=np.float64) * np.exp(float(c))
    assert np.max(np.abs(got - ref)) <= tol * np.max(terms)      # a sum of zero-mean products: accuracy relative to its terms
    t2, c2 = contract(einstr, *ops, split_format=True)
    assert np.array_equal(t, t2) and float(c) == float(c2)
    if einstr == "abc,abc->":                                        # fed by a rescaled intermediate: lazy rescale in the reduce pass
        w = (rng.standard_normal((64, 64)) * 3).astype(dtype)
        t3, c3 = contract("abd,dc,abc->", ops[0], w, ops[1], optimize=((0, 1), (0, 1)), split_format=True)
        ref3 = np.einsum("abd,dc,abc->", ops[0].astype(np.float64), w.astype(np.float64), ops[1].astype(np.float64))
        terms3 = np.einsum("abd,dc,abc->", *[np.abs(x).astype(np.float64) for x in (ops[0], w, ops[1])])
        assert abs(float(t3) * np.exp(float(c3)) - ref3) <= tol * terms3


@pytest.mark.parametrize("dtype,tol", [(np.float32, 2e-5), (np.float64, 1e-12)])
@pytest.mark.parametrize("einstr,shapes", [
    ("ab->b", [(2048, 512)]),                             # column sums: few outputs, long strided K
    ("ab,ab->b", [(3000, 260), (3000, 260)]),             # the same with a product; K % splits != 0, 65 vectors
    ("ab,cb->b", [(1500, 64), (7, 64)]),                  # K = (a, c) composite
    ("xab,xab->xb", [(2, 1200, 36), (2, 1200, 36)]),      # batch label
])
def test_streaming_step_with_few_outputs_and_long_k_is_split(dtype, tol, einstr, shapes):
    """A streaming (non-MFMA) step whose outputs fill only a few workgroups and whose K is long runs its K range
    split over workgroups, partial sums through the split-K reduce pass: same value, same bits run to run, and
    correct when an operand is a rescaled intermediate."""
    rng = np.random.default_rng(19)
    ops = [(rng.standard_normal(s) + 0.25).astype(dtype) for s in shapes]
    t, c = contract(einstr, *ops, split_format=True)
    ref = np.einsum(einstr, *[o.astype(np.float64) for o in ops])
    terms = np.einsum(einstr, *[np.abs(o).astype(np.float64) for o in ops])
    got = np.asarray(t, dtype=np.float64) * np.exp(float(c))
    assert np.max(np.abs(got - ref) / terms) <= tol
    t2, c2 = contract(einstr, *ops, split_format=True)
    assert np.array_equal(t, t2) and float(c) == float(c2)
    if einstr == "ab,ab->b":
        w = (rng.standard_normal((260, 260)) * 2).astype(dtype)
        t3, c3 = contract("ad,db,ab->b", ops[0], w, ops[1], optimize=((0, 1), (0, 1)), split_format=True)
        ref3 = np.einsum("ad,db,ab->b", *[x.astype(np.float64) for x in (ops[0], w, ops[1])])
        terms3 = np.einsum("ad,db,ab->b", *[np.abs(x).astype(np.float64) for x in (ops[0], w, ops[1])])
        assert np.max(np.abs(np.asarray(t3, dtype=np.float64) * np.exp(float(c3)) - ref3) / terms3) <= tol


@pytest.mark.parametrize("dtype,tol", [(np.float32, 2e-5), (np.float64, 1e-12)])
@pytest.mark.parametrize("einstr,shapes", [
    ("ab,b->a", [(100, 40000), (40000,)]),                # GEMV with few rows: 100 waves would each walk 40,000 terms
    ("ab,ab->a", [(300, 8200), (300, 8200)]),             # more than 64 workgroups' worth of outputs (collapse-mode partials)
    ("abc,bc->a", [(70, 64, 128), (64, 128)]),            # composite K
    ("xab,xb->xa", [(2, 90, 5000), (2, 5000)]),           # batch label
])
def test_row_dot_step_with_few_outputs_and_long_k_is_split(dtype, tol, einstr, shapes):
    """Row-dot steps (one wave per output, lanes along a unit-stride K) with too few outputs to fill the chip split
    their K range over workgroups as well (partial sums through the split-K reduce pass)."""
    rng = np.random.default_rng(23)
    ops = [(rng.standard_normal(s) + 0.25).astype(dtype) for s in shapes]
    t, c = contract(einstr, *ops, split_format=True)
    ref = np.einsum(einstr, *[o.astype(np.float64) for o in ops])
    terms = np.einsum(einstr, *[np.abs(o).astype(np.float64) for o in ops])
    got = np.asarray(t, dtype=np.float64) * np.exp(float(c))
    assert np.max(np.abs(got - ref) / terms) <= tol
    assert abs(np.mean(np.abs(t)) - 1.0) < (1e-5 if dtype == np.float32 else 1e-13)     # the abs-sum partials are right
    t2, c2 = contract(einstr, *ops, split_format=True)
    assert np.array_equal(t, t2) and float(c) == float(c2)


@pytest.mark.parametrize("dtype,tol", [(np.float32, 2e-5), (np.float64, 1e-12)])
@pytest.mark.parametrize("einstr,shapes", [
    ("abk,k->ab", [(37, 44, 16), (16,)]),                 # a vector applied to the innermost leg
    ("abk,ck->abc", [(20, 12, 8), (3, 8)]),               # a small matrix: output index c comes from the other operand
    ("xak,xk->xa", [(5, 300, 12), (5, 12)]),              # batch label
    ("abk,bk->ab", [(64, 40, 64), (40, 64)]),             # K = 64, second operand indexed by an output label
    ("akb,k->ab", [(33, 16, 8), (16,)]),                  # NOT this kernel (K strided in the operand): the generic path
])
def test_short_contiguous_k_under_a_strided_output_index(dtype, tol, einstr, shapes):
    """k_stream_kvec: same results as NumPy, and as a step fed by a rescaled intermediate."""
    rng = np.random.default_rng(29)
    ops = [(rng.standard_normal(s) + 0.25).astype(dtype) for s in shapes]
    t, c = contract(einstr, *ops, split_format=True)
    ref = np.einsum(einstr, *[o.astype(np.float64) for o in ops])
    terms = np.einsum(einstr, *[np.abs(o).astype(np.float64) for o in ops])
    got = np.asarray(t, dtype=np.float64) * np.exp(float(c))
    assert np.max(np.abs(got - ref) / terms) <= tol
    assert abs(np.mean(np.abs(t)) - 1.0) < (1e-5 if dtype == np.float32 else 1e-13)
    if einstr == "abk,k->ab":
        w = (rng.standard_normal((16, 16)) * 2).astype(dtype)
        t3, c3 = contract("abj,jk,k->ab", ops[0], w, ops[1], optimize=((0, 1), (0, 1)), split_format=True)
        ref3 = np.einsum("abj,jk,k->ab", *[x.astype(np.float64) for x in (ops[0], w, ops[1])])
        terms3 = np.einsum("abj,jk,k->ab", *[np.abs(x).astype(np.float64) for x in (ops[0], w, ops[1])])
        assert np.max(np.abs(np.asarray(t3, dtype=np.float64) * np.exp(float(c3)) - ref3) / terms3) <= tol


def test_k_split_forms_with_replicas_in_flight():
    """The K-split forms with three replicas per launch: every replica (different data) against NumPy."""
    cases = [("abc,abc->", [(64, 64, 64), (64, 64, 64)]), ("ab,b->a", [(100, 40000), (40000,)]),
             ("ab,ab->a", [(300, 8200), (300, 8200)]), ("ab->b", [(2048, 512)]),
             ("km,kn->mn", [(19200, 64), (19200, 64)]), ("ka,kb->ab", [(100000, 8), (100000, 8)])]
    for dtype, tol in ((np.float32, 2e-5), (np.float64, 1e-12)):
        for ein, shapes in cases:
            rng = np.random.default_rng(5)
            sets = [[(rng.standard_normal(s) + 0.25).astype(dtype) * (r + 1) for s in shapes] for r in range(3)]
            bc = E.BatchedContraction(ein, shapes, dtype, optimize=((0, 1),) if len(shapes) == 2 else ((0,),), replicas=3)
            t, c = bc.run_host(sets)
            bc.executor.close()
            for r in range(3):
                ref = np.einsum(ein, *[o.astype(np.float64) for o in sets[r]])
                terms = np.einsum(ein, *[np.abs(o).astype(np.float64) for o in sets[r]])
                got = np.asarray(t[r], dtype=np.float64) * np.exp(float(c[r]))
                assert np.max(np.abs(got - ref) / terms) <= tol, (ein, np.dtype(dtype).name, r)


# ---- extreme magnitudes: the lazy epilogue rescale must not change what the reference computes -----------------
def _plan_and_executor(g, dtype="float32", replicas=1):
    shapes = tuple(o.shape for o in g["operands"])
    clist = E._contract_path(g["einsum_str"], shapes, optimize=g["path"], memory_limit=None, use_blas=True)
    plan = E._native_plan(clist, shapes, dtype)
    return plan, engine.Executor(plan, replicas=replicas)


def test_huge_operands_switch_the_executor_to_eager_rescale():
    """fp32 operands of magnitude 1e13 (golden `mps_overlap_4x32x4_f32_huge`, generated by the reference): the
    reference normalises every intermediate BEFORE the next product and stays finite (log-value 252.28).  A tile
    kernel that accumulates on un-normalised operands overflows there (4e26 * 1e13 * sqrt(K) > 3.4e38); the executor
    must notice it from the scale registers, repeat the contraction with eager rescaling, and report the
    reference's value - never a silent inf."""
    g = load_golden("mps_overlap_4x32x4_f32_huge")
    plan, ex = _plan_and_executor(g)
    assert any(i["kernel"] == 2 for i in plan.step_infos()), "the fixture must exercise the MFMA tile kernels"
    outs, _log, resc = ex.run_host([g["operands"]])
    assert ex.eager_reruns() == 1, "overflow of the lazy rescale was not detected"
    assert np.all(np.isfinite(resc)) and np.all(np.isfinite(outs))
    c = E.accumulate_log_scale(resc[0], np.dtype(np.float32))
    assert float(outs[0]) == float(g["t_hat"]) and abs(float(c) - float(g["log_scale"])) <= 2e-5 * float(g["log_scale"])
    # every new run tries the lazy form first (graph replay, no renorm passes): the same extreme operands are caught
    # again and give the same bits; after three such runs in a row the executor stays eager
    for reruns in (2, 3, 3, 3):
        outs2, _log2, resc2 = ex.run_host([g["operands"]])
        assert ex.eager_reruns() == reruns
        np.testing.assert_array_equal(resc2, resc)
        np.testing.assert_array_equal(outs2, outs)
    # forcing eager mode from the start is the same computation
    plan2, ex2 = _plan_and_executor(g)
    assert ex2.set_rescale_mode(1) == 0
    outs3, _log3, resc3 = ex2.run_host([g["operands"]])
    assert ex2.eager_reruns() == 0
    np.testing.assert_array_equal(resc3, resc)
    np.testing.assert_array_equal(outs3, outs)


def test_ordinary_operands_after_an_overflowing_run_are_lazy_again():
    """One extreme input must not leave a cached executor in eager mode for the rest of the process (round-2 advice):
    the next, ordinary operands run lazily - no repeat, and bit-identical to an executor that never saw the
    extreme ones, also once the launch sequence is replayed as a hipGraph."""
    g = load_golden("mps_overlap_4x32x4_f32_huge")
    plan, ex = _plan_and_executor(g)
    _plan, fresh = _plan_and_executor(g)
    ex.run_host([g["operands"]])
    assert ex.eager_reruns() == 1
    tame = [(o / np.float32(1e13)).astype(np.float32) for o in g["operands"]]
    ref_o, _l, ref_r = fresh.run_host([tame])
    for _ in range(4):                       # eager launches, capture, replays
        o, _l2, r = ex.run_host([tame])
        assert ex.eager_reruns() == 1 and fresh.eager_reruns() == 0
        np.testing.assert_array_equal(o, ref_o)
        np.testing.assert_array_equal(r, ref_r)
    ex.close(); fresh.close()


def test_eager_and_lazy_rescale_agree_on_ordinary_data():
    """Eager mode is the reference's literal order of operations (T / s stored, then multiplied); on well-scaled
    data the default lazy mode must agree with it to rounding, and must not trigger a repeat."""
    for name, dtype, tol in (("mps_overlap_5x64x4_f32", "float32", 2e-6), ("mps_overlap_4x48x4_f64", "float64", 1e-14),
                             ("cp_r48_f32", "float32", 2e-6), ("peps3x3_D2_f32", "float32", 2e-6)):
        g = load_golden(name)
        plan, lazy = _plan_and_executor(g, dtype)
        _plan, eager = _plan_and_executor(g, dtype)
        eager.set_rescale_mode(1)
        o_l, _a, r_l = lazy.run_host([g["operands"]])
        o_e, _b, r_e = eager.run_host([g["operands"]])
        assert lazy.eager_reruns() == 0 and eager.eager_reruns() == 0
        assert rel_err(o_l, o_e) <= 10 * tol, name
        c_l = E.accumulate_log_scale(r_l[0], np.dtype(dtype))
        c_e = E.accumulate_log_scale(r_e[0], np.dtype(dtype))
        assert abs(float(c_l) - float(c_e)) <= tol * max(1.0, abs(float(c_e))) * 10, name
        # the eager result against the golden (reference) as well
        assert abs(float(c_e) - float(g["log_scale"])) <= (1e-10 if dtype == "float64" else 2e-5) * max(1.0, abs(float(g["log_scale"])))


def test_underflowing_network_matches_the_reference_zero():
    """Magnitude 1e-14: below the reference's min_norm threshold nothing is ever rescaled (einsum.py:94-102) and
    the product underflows to an exact zero in the reference itself; same here, register untouched."""
    g = load_golden("mps_overlap_4x32x4_f32_tiny")
    t, c = contract(g["einsum_str"], *g["operands"], optimize=g["path"], split_format=True)
    assert float(t) == 0.0 == float(g["t_hat"]) and float(c) == 0.0 == float(g["log_scale"])


# ---- fused steps: an element-wise product formed on the fly as the A operand of the GEMM that consumes it -----------
@pytest.fixture
def force_fusion(monkeypatch):
    """The planner fuses only where the intermediate is large (>= 2^20 elements); these tests force it on small
    networks (CTN_FUSE is read when a plan is created, so the plan cache is cleared around them)."""
    monkeypatch.setenv("CTN_FUSE", "1")
    E.clear_caches()
    yield
    E.clear_caches()


def _fused_infos(einstr, shapes, path):
    clist = E._contract_path(einstr, tuple(tuple(s) for s in shapes), optimize=path, memory_limit=None, use_blas=True)
    return E._native_plan(clist, tuple(tuple(s) for s in shapes), "float32").step_infos()


@pytest.mark.parametrize("einstr,shapes,path", [
    ("ac,ad,ae->cde", [(48, 40), (48, 36), (48, 44)], ((0, 1), (0, 1))),          # CP: Khatri-Rao then GEMM (pattern A)
    ("ac,ad,ae->cde", [(130, 33), (130, 20), (130, 70)], ((0, 1), (0, 1))),       # ragged everything, K = 130
    ("xac,xad,xae->xcde", [(2, 64, 16), (2, 64, 24), (2, 64, 40)], ((0, 1), (0, 1))),  # a batch label through all three
    ("ac,ad,ae->edc", [(64, 32), (64, 16), (64, 48)], ((0, 1), (0, 1))),          # last step with M innermost in the output
    ("m,mk,kn->mn", [(256,), (256, 64), (64, 96)], ((0, 1), (0, 1))),             # a vector scaling the rows (Hadamard, pattern A)
])
def test_fused_product_as_gemm_operand_vs_numpy(einstr, shapes, path, force_fusion):
    rng = np.random.default_rng(17)
    ops = [(rng.standard_normal(s) * rng.uniform(0.5, 2.0)).astype(np.float32) for s in shapes]
    infos = _fused_infos(einstr, shapes, path)
    assert [i["kernel"] for i in infos] == [5, 2] and infos[1]["mode_a"] >= 3, infos
    t_hat, c = contract(einstr, *ops, optimize=path, split_format=True)
    ref = np.einsum(einstr, *[o.astype(np.float64) for o in ops])
    got = t_hat.astype(np.float64) * np.exp(float(c))
    assert got.shape == ref.shape
    assert np.max(np.abs(got - ref)) <= 2e-4 * np.max(np.abs(ref))
    assert abs(np.mean(np.abs(t_hat)) - 1.0) < 1e-5


@pytest.fixture
def force_fusion_kr(monkeypatch):
    """CTN_FUSE=2: every Khatri-Rao pattern forced, the epilogue-sum pattern (which would take the batched-MPS sites
    first) off."""
    monkeypatch.setenv("CTN_FUSE", "2")
    E.clear_caches()
    yield
    E.clear_caches()


def test_fused_gemm_with_reweighting_consumer_batched_mps(force_fusion_kr):
    """Pattern B: an MPS site applied to a batch of inputs - `bl,plr->bpr` then `bpr,bp->br` - runs as ONE GEMM over
    (p, l) whose A operand v[b,l] * x[b,p] is formed on the fly; v is a rescaled intermediate of the previous site.
    Against the oracle on the same path (the oracle materialises every intermediate)."""
    from contractn_amd import TN
    from contractn_amd.paths import ssa_to_linear
    from oracle import cpu_ref
    from tests import networks as nets

    B, n_sites, bond, phys = 256, 6, 64, 4
    tn, inputs = nets.batched_mps(TN, n_sites, bond, phys, B, dtype=np.float32, seed=4)
    path = ssa_to_linear(nets.batched_mps_path(n_sites), 2 * n_sites)
    ops = E.make_arg_packer(tn)(tn.params, inputs)
    shapes = [o.shape for o in ops]
    infos = _fused_infos(tn.einsum_str, shapes, path)
    assert sum(i["kernel"] == 5 for i in infos) == n_sites - 2 and sum(i["mode_a"] >= 3 for i in infos) == n_sites - 2
    assert max(i["out_numel"] for i in infos if i["kernel"] != 5) <= B * bond      # the B x phys x bond tensor never exists
    fun = tn.make_contract_fun(optimize=path, split_format=True)
    t, c = fun(tn.params, inputs)
    clist = E._contract_path(tn.einsum_str, tuple(shapes), optimize=path, memory_limit=None, use_blas=True)
    rt, rc, _ = cpu_ref.core_contract(list(ops), clist)
    got = t.astype(np.float64) * np.exp(float(c))
    ref = np.asarray(rt, dtype=np.float64) * np.exp(float(rc))
    assert np.max(np.abs(got - ref)) <= 1e-3 * np.max(np.abs(ref))
    # replicas in flight and the unfused plan give the same numbers to rounding
    t2, c2 = fun(tn.params, inputs)
    assert np.array_equal(t2, t) and float(c2) == float(c)


# ---- pattern C: a short label re-weighted by a network input and summed in the GEMM's epilogue ----------------------
@pytest.mark.parametrize("einstr,shapes", [
    ("bl,plr,bp->br", [(256, 64), (4, 64, 64), (256, 4)]),          # one batched-MPS site, d = 4
    ("bl,plr,bp->br", [(1200, 40), (2, 40, 33), (1200, 2)]),        # d = 2, ragged rows, N = 66 (half a vector at the edge)
    ("bl,plr,bp->br", [(390, 24), (4, 24, 50), (390, 4)]),          # d = 4, ragged rows and columns
    ("bl,lrp,bp->br", [(256, 32), (32, 64, 4), (256, 4)]),          # the short label innermost in the core too
    ("xbl,xplr,bp->xbr", [(3, 128, 16), (3, 4, 16, 64), (128, 4)]), # a batch label the weights do not carry
    ("bl,plr,bp->rb", [(256, 64), (4, 64, 64), (256, 4)]),          # last step, caller's axis order
    ("abl,plr,abp->abr", [(16, 32, 48), (2, 48, 64), (16, 32, 2)]), # two row labels in the weights
    ("bl,plr,bp->br", [(192, 64), (4, 64, 96), (192, 4)]),          # 64-row tiles (M = 192)
    ("abl,plr,bp->abr", [(16, 32, 48), (4, 48, 32), (32, 4)]),     # a row label the weights do not carry
])
def test_epilogue_sum_step_vs_numpy(einstr, shapes):
    rng = np.random.default_rng(23)
    ops = [(rng.standard_normal(s) * rng.uniform(0.5, 2.0)).astype(np.float32) for s in shapes]
    path = ((0, 1), (0, 1))
    E.clear_caches()
    infos = _fused_infos(einstr, shapes, path)
    ext_p = shapes[2][-1]
    assert [i["kernel"] for i in infos] == [5, 2] and infos[1]["epilogue_sum"] == ext_p, infos
    assert infos[1]["out_numel"] * ext_p == infos[1]["batch"] * infos[1]["m"] * infos[1]["n"]
    t_hat, c = contract(einstr, *ops, optimize=path, split_format=True)
    ref = np.einsum(einstr, *[o.astype(np.float64) for o in ops])
    got = t_hat.astype(np.float64) * np.exp(float(c))
    assert got.shape == ref.shape
    assert np.max(np.abs(got - ref)) <= 2e-4 * np.max(np.abs(ref))
    assert abs(np.mean(np.abs(t_hat)) - 1.0) < 1e-5
    E.clear_caches()


def test_epilogue_sum_batched_mps_vs_oracle_and_vs_unfused(monkeypatch):
    """The paper's ML workload (reference README, Fig. 1d): every interior site of a batched MPS evaluation is ONE
    launch - `bl,plr->bpr` with `bpr,bp->br` folded into its epilogue; the (B, d, D) tensor never exists.  Against the
    oracle on the same path, against the unfused plan, with replicas in flight, and twice for bit-identity."""
    from contractn_amd import TN
    from contractn_amd.paths import ssa_to_linear
    from oracle import cpu_ref
    from tests import networks as nets

    monkeypatch.delenv("CTN_FUSE", raising=False)
    E.clear_caches()
    B, n_sites, bond, phys = 256, 6, 64, 4
    tn, inputs = nets.batched_mps(TN, n_sites, bond, phys, B, dtype=np.float32, seed=4)
    path = ssa_to_linear(nets.batched_mps_path(n_sites), 2 * n_sites)
    ops = E.make_arg_packer(tn)(tn.params, inputs)
    shapes = [o.shape for o in ops]
    infos = _fused_infos(tn.einsum_str, shapes, path)
    assert sum(i["kernel"] == 5 for i in infos) == n_sites - 2 == sum(i["epilogue_sum"] == phys for i in infos), infos
    assert max(i["out_numel"] for i in infos if i["kernel"] != 5) <= B * bond
    fun = tn.make_contract_fun(optimize=path, split_format=True)
    t, c = fun(tn.params, inputs)
    clist = E._contract_path(tn.einsum_str, tuple(shapes), optimize=path, memory_limit=None, use_blas=True)
    rt, rc, _ = cpu_ref.core_contract(list(ops), clist)
    got = t.astype(np.float64) * np.exp(float(c))
    ref = np.asarray(rt, dtype=np.float64) * np.exp(float(rc))
    assert np.max(np.abs(got - ref)) <= 1e-3 * np.max(np.abs(ref))
    t2, c2 = fun(tn.params, inputs)
    assert np.array_equal(t2, t) and float(c2) == float(c)
    # three replicas of the plan in one launch: replica 0 carries the same operands
    bc = E.BatchedContraction(tn.einsum_str, shapes, np.float32, optimize=path, replicas=3)
    rng = np.random.default_rng(5)
    reps = [list(ops)] + [[(o * rng.uniform(0.5, 1.5)).astype(np.float32) for o in ops] for _ in range(2)]
    tb, cb = bc.run_host(reps)
    assert np.array_equal(tb[0], t) and float(cb[0]) == float(c)
    # the unfused plan: same numbers to rounding
    monkeypatch.setenv("CTN_FUSE", "0")
    E.clear_caches()
    assert all(i["kernel"] != 5 for i in _fused_infos(tn.einsum_str, shapes, path))
    t0, c0 = tn.make_contract_fun(optimize=path, split_format=True)(tn.params, inputs)
    un = t0.astype(np.float64) * np.exp(float(c0))
    assert np.max(np.abs(got - un)) <= 1e-5 * np.max(np.abs(un))
    E.clear_caches()


def test_fusion_is_off_for_small_intermediates_and_switchable(monkeypatch):
    monkeypatch.delenv("CTN_FUSE", raising=False)
    E.clear_caches()
    small = _fused_infos("ac,ad,ae->cde", [(48, 40), (48, 36), (48, 44)], ((0, 1), (0, 1)))
    assert all(i["kernel"] != 5 for i in small)                               # 69k-element product: kept as a step
    mid = _fused_infos("ac,ad,ae->cde", [(1024, 1024), (1024, 1024), (1024, 512)], ((0, 1), (0, 1)))
    assert all(i["kernel"] != 5 for i in mid)                                 # 2^30-element product (4 GiB): materialised (19.6 vs 22.5 ms)
    wide = _fused_infos("ac,ad,ae->cde", [(4096, 1024), (4096, 1024), (4096, 1024)], ((0, 1), (0, 1)))
    assert all(i["kernel"] != 5 for i in wide) and wide[1]["batch"] == 1024   # 2^32-element product: materialised too (66.6 vs 100.7 ms),
    #                                                                          laid out so that its consumer sums an inner label
    big = _fused_infos("ac,ad,ae->cde", [(8192, 1024), (8192, 1024), (8192, 512)], ((0, 1), (0, 1)))
    assert [i["kernel"] for i in big] == [5, 2]                              # 2^33-element product (32 GiB): fused by default
    monkeypatch.setenv("CTN_FUSE", "0")
    E.clear_caches()
    off = _fused_infos("ac,ad,ae->cde", [(1024, 1024), (1024, 1024), (1024, 512)], ((0, 1), (0, 1)))
    assert all(i["kernel"] != 5 for i in off)
    monkeypatch.setenv("CTN_FUSE", "1")
    E.clear_caches()
    on = _fused_infos("ac,ad,ae->cde", [(1024, 1024), (1024, 1024), (1024, 512)], ((0, 1), (0, 1)))
    assert [i["kernel"] for i in on] == [5, 2]
    monkeypatch.delenv("CTN_FUSE")
    E.clear_caches()


# ---- leaf steps hoisted and sent out as one launch ------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_grouped_leaf_steps_give_the_same_bits_as_one_launch_each(dtype, monkeypatch):
    """A PEPS path that absorbs each physical vector right before its site joins the boundary: the engine moves the
    absorptions (steps on two network inputs) to the front and launches them as ONE kernel.  Same arithmetic per
    step and the register still accumulated in the caller's step order, so the result is bit-identical to the plain
    order with one launch per step (CTN_HOIST=0 CTN_GROUP=0) - and equal to the oracle on the caller's path."""
    from contractn_amd import TN
    from contractn_amd.paths import ssa_to_linear
    from oracle import cpu_ref
    from tests import networks as nets
    from tests.test_plan_host import _interleaved_peps_path

    rows, cols, bond = 4, 5, 4
    tn = nets.peps_closed(TN, rows, cols, bond, dtype=dtype, seed=6)
    path = ssa_to_linear(_interleaved_peps_path(rows, cols), 2 * rows * cols)
    results = {}
    for mode, env in (("grouped", {"CTN_HOIST": "1", "CTN_GROUP": "1"}), ("plain", {"CTN_HOIST": "0", "CTN_GROUP": "0"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        E.clear_caches()
        fun = tn.make_contract_fun(optimize=path, split_format=True)
        outs = [fun(tn.params, ()) for _ in range(4)]           # eager launches, then the captured graph
        assert all(float(t) == float(outs[0][0]) and float(c) == float(outs[0][1]) for t, c in outs)
        results[mode] = (float(outs[0][0]), float(outs[0][1]))
    E.clear_caches()
    assert results["grouped"] == results["plain"]
    rt, rc = cpu_ref.contract(tn.einsum_str, *tn.params, path=path, split_format=True)
    tol = 1e-4 if dtype == np.float32 else 1e-10
    assert results["grouped"][0] == float(rt) and abs(results["grouped"][1] - float(rc)) <= tol * max(1.0, abs(float(rc)))


def test_leaf_group_arguments_built_inside_the_graph_capture(monkeypatch):
    """Event-timed enqueues send every step out on its own, so when the first two enqueues of an executor are timed
    the leaf group's arguments are built (and uploaded from pinned memory) by the THIRD one - the one that is being
    captured into the hipGraph.  Replays must still give the ungrouped bits."""
    import torch

    from contractn_amd import TN
    from contractn_amd.paths import ssa_to_linear
    from tests import networks as nets
    from tests.test_plan_host import _interleaved_peps_path

    rows, cols, bond = 3, 4, 4
    tn = nets.peps_closed(TN, rows, cols, bond, dtype=np.float32, seed=8)
    path = ssa_to_linear(_interleaved_peps_path(rows, cols), 2 * rows * cols)
    shapes = [p.shape for p in tn.params]
    dev_ops = [torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in tn.params]
    got = {}
    for mode, group in (("grouped", "1"), ("plain", "0")):
        monkeypatch.setenv("CTN_GROUP", group)
        E.clear_caches()
        bc = E.BatchedContraction(tn.einsum_str, shapes, np.float32, optimize=path, replicas=1)
        out = torch.zeros((1,) + tuple(bc.plan.out_shape), device="cuda")
        launch = bc.executor.make_enqueue([t.data_ptr() for t in dev_ops], [out[0].data_ptr()])
        bc.executor.set_timing(2)
        vals = []
        for _ in range(6):
            launch()
            c = bc.fetch_log_scale()
            vals.append((float(out[0].cpu()), float(c[0])))
        assert all(v == vals[0] for v in vals), vals
        got[mode] = vals[0]
        del bc
    E.clear_caches()
    assert got["grouped"] == got["plain"]


def test_epilogue_sum_and_grouped_leaves_in_eager_rescale_mode():
    """Eager mode normalises every intermediate in place right after its step: an epilogue-summed step must then see
    scale 1 on its operands and have ITS output (the summed one) renormalised; leaf groups fall back to one launch
    per step.  Same numbers as the lazy default, to rounding - a batched MPS (pattern C) and an interleaved PEPS."""
    from contractn_amd import TN, engine
    from contractn_amd.paths import ssa_to_linear
    from tests import networks as nets
    from tests.test_plan_host import _interleaved_peps_path

    E.clear_caches()
    tn, inputs = nets.batched_mps(TN, 6, 64, 4, 512, dtype=np.float32, seed=4)
    ops = E.make_arg_packer(tn)(tn.params, inputs)
    cases = [(tn.einsum_str, [np.asarray(o) for o in ops], ssa_to_linear(nets.batched_mps_path(6), 12))]
    tp = nets.peps_closed(TN, 3, 4, 4, dtype=np.float32, seed=6)
    cases.append((tp.einsum_str, list(tp.params), ssa_to_linear(_interleaved_peps_path(3, 4), 24)))
    for einstr, operands, path in cases:
        shapes = tuple(o.shape for o in operands)
        clist = E._contract_path(einstr, shapes, optimize=path, memory_limit=None, use_blas=True)
        plan = E._native_plan(clist, shapes, "float32")
        lazy, eager = engine.Executor(plan), engine.Executor(plan)
        eager.set_rescale_mode(1)
        for ex in (lazy, eager, eager):                      # (the second eager run: nothing is left over from the first)
            o, _l, r = ex.run_host([operands])
            c = float(E.accumulate_log_scale(r[0], np.dtype(np.float32)))
            val = o[0].astype(np.float64) * np.exp(c)
            if ex is lazy:
                ref = val
        assert lazy.eager_reruns() == 0 and eager.eager_reruns() == 0
        assert np.max(np.abs(val - ref)) <= 2e-5 * np.max(np.abs(ref)), einstr
        lazy.close(); eager.close()
    E.clear_caches()


# ---- torch backend: the register has the tensor dtype (reference einsum.py:338) -------------------
@pytest.mark.parametrize("name", golden_names(torch_backend=True))
@pytest.mark.parametrize("where", ["cpu", "cuda"])
def test_torch_operands_keep_the_register_in_fp32(name, where):
    """Fixtures: the unmodified reference on torch CPU tensors.  fp32 torch operands give a ``torch.float32`` 0-d
    register accumulated step by step in fp32 - 1000 x log(3) = 1098.6213 (bits 0x1.12a7c4p+10), where NumPy
    operands give the float64 register 1098.6123 - for host and device tensors alike."""
    import torch

    g = load_golden(name)
    ops = [torch.from_numpy(o).to(where) for o in g["operands"]]
    t_hat, log_scale = contract(g["einsum_str"], *ops, optimize=g["path"], split_format=True)
    assert isinstance(log_scale, torch.Tensor) and log_scale.dtype == torch.float32 and log_scale.dim() == 0
    assert t_hat.dtype == torch.float32 and t_hat.device.type == where and log_scale.device.type == where
    assert rel_err(t_hat.cpu().numpy(), g["t_hat"]) <= 2e-5
    if name.startswith("readme_chain1000"):
        assert float(log_scale).hex() == g["log_scale_hex"] == "0x1.12a7c40000000p+10"
        t_np, c_np = contract(g["einsum_str"], *g["operands"], optimize=g["path"], split_format=True)
        # the NumPy register differs: fp32 logs added up in float64 (SURVEY.md App. A, row float32)
        assert c_np.dtype == np.float64 and abs(float(c_np) - 1000 * float(np.log(np.float32(3.0)))) < 1e-9
    else:
        assert abs(float(log_scale) - float(g["log_scale"])) <= 2e-5 * max(1.0, abs(float(g["log_scale"])))
    out = contract(g["einsum_str"], *ops, optimize=g["path"])
    assert out.dtype == torch.float32                                  # torch: fp32 x fp32 register stays fp32


# ---- steps on more than two operands (reference einsum.py:382-384 hands any step to _einsum) -------------------------
def test_nary_steps_and_optimize_false():
    """`optimize=False` is ONE einsum over all operands in the reference; an explicit path may also name three or
    more operands per step.  The engine orders such a step pairwise (greedy on the sub-network): same value as
    np.einsum, including hyperedges, a trace and a label summed out of a single operand inside the n-ary step."""
    rng = np.random.default_rng(21)
    cases = [
        ("ab,bc,cd,de->ae", [(30, 40), (40, 50), (50, 60), (60, 70)], False),
        ("a,ab,abc,cdd,e->ae", [(3,), (3, 4), (3, 4, 5), (5, 2, 2), (6,)], ((0, 1, 2), (0, 1, 2))),
        ("ab,ab,ab->ab", [(9, 7)] * 3, False),
        ("ijk,jl,km,lmn->in", [(8, 9, 10), (9, 11), (10, 12), (11, 12, 13)], ((0, 1, 2, 3),)),
    ]
    for ein, shapes, opt in cases:
        for dtype, tol in ((np.float64, 1e-12), (np.float32, 2e-5)):
            ops = [rng.standard_normal(sh).astype(dtype) for sh in shapes]
            ref = np.einsum(ein, *[o.astype(np.float64) for o in ops])
            t, c = contract(ein, *ops, optimize=opt, split_format=True)
            got = np.asarray(t, dtype=np.float64) * np.exp(float(c))
            assert rel_err(got, ref) <= tol, (ein, np.dtype(dtype).name)
            assert abs(np.mean(np.abs(t)) - 1.0) < 1e-5


# ---- the one-tile-per-CU form (k_mfma_f32_h): 128 x 128 tiles, K split over the two halves of an 8-wave workgroup ---------
@pytest.fixture
def force_h_form(monkeypatch):
    """The launcher takes this form when a step is about one 128 x 128 tile per CU; these tests force it for every
    eligible step (operands in 16-byte-request modes, K a multiple of 32) - and keep chains of epilogue-summed steps
    on their per-site launches (no k_sweep_f32)."""
    monkeypatch.setenv("CTN_H", "1")
    monkeypatch.setenv("CTN_SWEEP", "0")
    E.clear_caches()
    yield
    E.clear_caches()


@pytest.mark.parametrize("einstr,shapes,path", [
    ("km,kn,nj->mj", [(64, 128), (64, 128), (128, 8)], ((0, 1), (0, 1))),        # one tile, two k-tiles per K half
    ("km,kn,nj->mj", [(96, 384), (96, 256), (256, 8)], ((0, 1), (0, 1))),        # 3 x 2 tiles, three k-tiles per half
    ("km,kn,nj->mj", [(544, 128), (544, 128), (128, 8)], ((0, 1), (0, 1))),      # 17 k-tiles per half: the ring wraps
    ("xkm,xkn,xnj->xmj", [(3, 64, 128), (3, 64, 128), (3, 128, 4)], ((0, 1), (0, 1))),   # batch (hyperedge) label
    ("km,kn,nj->mj", [(64, 200), (64, 100), (100, 8)], ((0, 1), (0, 1))),        # ragged M and N: masked edge tiles
    ("mk,kn,nj->mj", [(256, 64), (64, 128), (128, 8)], ((0, 1), (0, 1))),        # row-major A: requests along k
    ("km,nk,nj->mj", [(96, 256), (384, 96), (384, 8)], ((0, 1), (0, 1))),        # k-contiguous B
    ("mk,nk,nj->mj", [(200, 128), (100, 128), (100, 8)], ((0, 1), (0, 1))),      # both, ragged M and N
    ("km,kn,nj,mj->", [(64, 256), (64, 256), (256, 16), (256, 16)], ((0, 1), (0, 1), (0, 1))),  # feeds rescaled intermediates on
])
def test_h_form_kernel_vs_numpy(einstr, shapes, path, force_h_form):
    rng = np.random.default_rng(17)
    ops = [(rng.standard_normal(s) * rng.uniform(0.5, 3.0)).astype(np.float32) for s in shapes]
    bc = E.BatchedContraction(einstr, shapes, np.float32, optimize=path, replicas=2)
    t, c = bc.run_host([ops, [2 * o for o in ops]])
    assert bc.executor.step_tiles()[0] == (128, 128) and bc.plan.step_infos()[0]["kernel"] == 2
    ref = np.einsum(einstr, *[o.astype(np.float64) for o in ops])
    for r, f in ((0, 1.0), (1, 2.0 ** len(ops))):
        got = np.asarray(t[r], dtype=np.float64) * np.exp(float(c[r]))
        assert np.max(np.abs(got - f * ref)) <= 1e-4 * f * np.max(np.abs(ref))
    t2, c2 = bc.run_host([ops, [2 * o for o in ops]])
    assert np.array_equal(t2, t) and np.array_equal(c2, c)           # fixed-order sums: the same bits
    bc.executor.close()


@pytest.mark.parametrize("phys", [2, 4])
def test_h_form_epilogue_summed_sites(phys, force_h_form):
    """The batched-MPS site step (`bl,plr->bpr` + `bpr,bp->br` as ONE launch: GEMM with the physical leg re-weighted
    and summed in the epilogue) on the one-tile-per-CU form, d = 2 and d = 4, ragged batch: against the oracle."""
    from contractn_amd import TN
    from contractn_amd.paths import ssa_to_linear
    from oracle import cpu_ref
    from tests import networks as nets

    tn, inputs = nets.batched_mps(TN, 5, 64, phys, 600, dtype=np.float32, seed=4)   # (600 x phys x 64 >= 2^16: fused by default)
    ops = [np.asarray(o) for o in E.make_arg_packer(tn)(tn.params, inputs)]
    path = ssa_to_linear(nets.batched_mps_path(5), 10)
    clist = E._contract_path(tn.einsum_str, tuple(o.shape for o in ops), optimize=path, memory_limit=None, use_blas=True)
    plan = E._native_plan(clist, tuple(o.shape for o in ops), "float32")
    assert sum(i["epilogue_sum"] == phys for i in plan.step_infos()) == 3
    ex = engine.Executor(plan)
    outs, _log, resc = ex.run_host([ops])
    tiles = ex.step_tiles()
    assert sum(tl == (128, 128) and i["epilogue_sum"] == phys for tl, i in zip(tiles, plan.step_infos())) == 3, tiles
    c = E.accumulate_log_scale(resc[0], np.dtype(np.float32))
    rt, rc = cpu_ref.contract(tn.einsum_str, *ops, path=list(path), split_format=True)
    got = outs[0].astype(np.float64) * np.exp(float(c))
    ref = np.asarray(rt, dtype=np.float64) * np.exp(float(rc))
    assert np.max(np.abs(got - ref)) <= 1e-4 * np.max(np.abs(ref))
    ex.close()


# ---- complex tensors (the reference contracts them through NumPy: modulus norm, real register) ------
@pytest.mark.parametrize("name", ["mps_overlap_5x12x3_c128", "mps_overlap_4x40x4_c64", "mps_overlap_4x10x3_mixed_c128",
                                  "mps_open_random_c128", "cp_r5_c128"])
def test_complex_networks_match_the_reference(name):
    """Fixtures from the unmodified reference on complex NumPy arrays.  The engine computes on (re, im) components
    (`einsum._complex_plan_cached`: complex x complex steps through the structure tensor of complex multiplication,
    real x complex steps as they are) and brings the result to the reference's normalisation: mean MODULUS one, the
    phase kept, a real float64 register - also for torch operands and for the de-stabilised product."""
    import torch

    g = load_golden(name)
    ops = g["operands"]
    c64 = g["t_hat"].dtype == np.complex64
    tol = 2e-5 if c64 else 1e-11
    t, c = contract(g["einsum_str"], *ops, optimize=g["path"], split_format=True)
    assert t.dtype == g["t_hat"].dtype and t.shape == g["t_hat"].shape and c.dtype == np.float64
    assert abs(np.mean(np.abs(t)) - 1.0) <= 10 * tol
    assert np.max(np.abs(t - g["t_hat"])) <= tol * max(1.0, float(np.max(np.abs(g["t_hat"]))))
    assert abs(float(c) - float(g["log_scale"])) <= tol * max(1.0, abs(float(g["log_scale"])))
    plain = contract(g["einsum_str"], *ops, optimize=g["path"])
    assert plain.dtype == g["plain"].dtype
    assert np.max(np.abs(plain - g["plain"])) <= 10 * tol * float(np.max(np.abs(g["plain"])))
    t_ops = [torch.from_numpy(o) for o in ops]
    tt, tc = contract(g["einsum_str"], *t_ops, optimize=g["path"], split_format=True)
    assert tt.dtype == (torch.complex64 if c64 else torch.complex128) and not tc.dtype.is_complex
    assert np.max(np.abs(tt.numpy() - g["t_hat"])) <= 10 * tol * max(1.0, float(np.max(np.abs(g["t_hat"]))))


# ---- zipper pairs as one launch (k_zip_f32): the intermediate of two consecutive GEMM steps stays in registers ------------
@pytest.mark.parametrize("sites,phys,replicas,form", [(4, 4, 2, "1"), (5, 2, 3, "1"), (6, 1, 1, "1"),
                                                      (5, 4, 2, "2"), (4, 2, 3, "2"), (6, 1, 2, "2")])
def test_zipper_pairs_in_one_launch_match_the_two_step_path(sites, phys, replicas, form, monkeypatch):
    """<phi|psi> of two MPS with bond 256 (the metric's network in small): with CTN_ZIP=1 every (E . psi_i, T . phi_i)
    pair of the zipper runs as ONE k_zip_f32 launch - T only ever exists in accumulators - against the two-launch
    path (CTN_ZIP=0) and the oracle; the first step of a pair reports rescale 0 (its magnitude moves into the second),
    the log-value is the same; physical dimension 4, 2 and 1 (q = 4, 2, 1 passes over the register-resident T).
    CTN_ZIP=2: the same through k_zip64_f32 (64 values of u per workgroup, four quarters of m1: the form 64 ... 127 networks
    in flight take by default)."""
    from contractn_amd import TN
    from contractn_amd.paths import ssa_to_linear
    from oracle import cpu_ref
    from tests import networks as nets

    tn, ssa = nets.mps_overlap(TN, sites, 256, phys, dtype=np.float32, seed=3, scale=16.0)
    path = ssa_to_linear(ssa, 2 * sites)
    shapes = [p.shape for p in tn.params]
    rng = np.random.default_rng(5)
    sets = [[(rng.standard_normal(sh) / 16.0).astype(np.float32) for sh in shapes] for _ in range(replicas)]
    res = {}
    monkeypatch.setenv("CTN_ZIPL", "0")                # (the latency form of the pairs has its own test below)
    for mode in ("0", form):
        monkeypatch.setenv("CTN_ZIP", mode)
        E.clear_caches()
        bc = E.BatchedContraction(tn.einsum_str, shapes, np.float32, optimize=path, replicas=replicas)
        t, c = bc.run_host(sets)
        t2, c2 = bc.run_host(sets)                     # graph capture / replay: the same bits
        assert np.array_equal(t, t2) and np.array_equal(c, c2)
        res["1" if mode == form else "0"] = (t, c, bc.executor.step_tiles(), bc.executor.fetch()[1])
        bc.executor.close()
    monkeypatch.delenv("CTN_ZIPL")
    tiles = res["1"][2]
    pairs = [s for s, tl in enumerate(tiles) if tl == ((512, 256) if form == "1" else (512, 128))]
    # every interior site but the first (whose E comes out of the opening step with the other leg innermost)
    assert len(pairs) == sites - 3 and all(tiles[s - 1] == (1, 1) for s in pairs), tiles
    assert all(res["1"][3][r][s - 1] == 0.0 for s in pairs for r in range(replicas))          # T is never rescaled
    assert not any(tl in ((512, 256), (512, 128), (1, 1)) for tl in res["0"][2])
    for r in range(replicas):
        rt, rc = cpu_ref.contract(tn.einsum_str, *sets[r], path=path, split_format=True)
        for mode in ("0", "1"):
            assert float(res[mode][0][r]) == float(rt) and abs(float(res[mode][1][r]) - float(rc)) <= 1e-4, (mode, r)
    monkeypatch.delenv("CTN_ZIP")
    E.clear_caches()


# ---- zipper pairs in their latency form (k_zip_lat): one launch per site for ONE network in flight ------------------------
@pytest.mark.parametrize("sites,phys,replicas,mp", [(5, 4, 1, 32), (8, 4, 3, 64), (7, 2, 2, 64), (4, 4, 1, 64), (6, 4, 2, 32)])
def test_zipper_pairs_latency_form_matches_the_two_step_path(sites, phys, replicas, mp, monkeypatch):
    """<phi|psi> of two MPS with bond 256 and ONE network (or a few) in flight - what `tn.contract()` itself runs: with
    CTN_ZIPL=1 every (E . psi_i, T . phi_i) pair is ONE k_zip_lat launch, cut over u AND m1; a pair's partial results leave
    as 8 (d = 4) or 4 (d = 2) slabs that the next pair adds up while loading them, the last pair's by k_zip_slab_sum.
    Against the per-step launches (CTN_ZIPL=0) and the oracle; the same bits on every repeat (eager launches, graph
    capture, replay); the first step of a pair reports rescale 0; the eager rescale mode (every pair followed by slab sum
    and renorm) and operands 1e9 times larger (the lazy guard) give the oracle's value too."""
    from contractn_amd import TN
    from contractn_amd.paths import ssa_to_linear
    from oracle import cpu_ref
    from tests import networks as nets

    tn, ssa = nets.mps_overlap(TN, sites, 256, phys, dtype=np.float32, seed=3, scale=16.0)
    path = ssa_to_linear(ssa, 2 * sites)
    shapes = [p.shape for p in tn.params]
    rng = np.random.default_rng(5)
    sets = [[(rng.standard_normal(sh) / 16.0).astype(np.float32) for sh in shapes] for _ in range(replicas)]
    res = {}
    monkeypatch.setenv("CTN_ZIP", "0")
    monkeypatch.setenv("CTN_ZIPL_MP", str(mp))         # the part of m1 per workgroup: 32 (8 slabs) or 64 (4 slabs)
    for mode in ("0", "1"):
        monkeypatch.setenv("CTN_ZIPL", mode)
        E.clear_caches()
        bc = E.BatchedContraction(tn.einsum_str, shapes, np.float32, optimize=path, replicas=replicas)
        t, c = bc.run_host(sets)
        for _ in range(3):                             # graph capture / replay: the same bits
            t2, c2 = bc.run_host(sets)
            assert np.array_equal(t, t2) and np.array_equal(c, c2)
        res[mode] = (t, c, bc.executor.step_tiles(), bc.executor.fetch()[1])
        if mode == "1":
            bc.executor.set_rescale_mode(1)            # eager: slab sum + renorm behind every pair, plain E in
            te, ce = bc.run_host(sets)
            bc.executor.set_rescale_mode(0)
            huge = [[(o * np.float32(3e8)).astype(np.float32) for o in ops] for ops in sets]
            th, ch = bc.run_host(huge)                 # the lazy guard may or may not have to repeat it eagerly
            t3, c3 = bc.run_host(sets)                 # ... and the next tame operands give the first bits again
            assert np.array_equal(t3, t) and np.array_equal(c3, c)
        bc.executor.close()
    monkeypatch.delenv("CTN_ZIPL")
    monkeypatch.delenv("CTN_ZIPL_MP")
    monkeypatch.delenv("CTN_ZIP")
    E.clear_caches()
    tiles = res["1"][2]
    tile = (mp, 256)
    pairs = [s for s, tl in enumerate(tiles) if tl == tile]
    assert len(pairs) == sites - 3 and all(tiles[s - 1] == (1, 1) for s in pairs), tiles
    assert all(res["1"][3][r][s - 1] == 0.0 for s in pairs for r in range(replicas))
    assert not any(tl in (tile, (1, 1)) for tl in res["0"][2])
    for r in range(replicas):
        rt, rc = cpu_ref.contract(tn.einsum_str, *sets[r], path=path, split_format=True)
        for mode in ("0", "1"):
            assert float(res[mode][0][r]) == float(rt) and abs(float(res[mode][1][r]) - float(rc)) <= 1e-4, (mode, r)
        assert float(te[r]) == float(rt) and abs(float(ce[r]) - float(rc)) <= 1e-4
        ht, hc = cpu_ref.contract(tn.einsum_str, *huge[r], path=path, split_format=True)
        assert float(th[r]) == float(ht) and abs(float(ch[r]) - float(hc)) <= 1e-3


@pytest.mark.parametrize("einstr,shapes", [
    ("ab,bc,cd->ad", [(512, 4096), (4096, 512), (512, 8)]),        # A k-contiguous, B row-contiguous: <4,2,asm,2,1>
    ("ba,bc,cd->ad", [(2048, 1000), (2048, 512), (512, 8)]),       # both row-contiguous, a ragged edge: <4,2,asm,1,1>
    ("ab,cb,cd->ad", [(256, 8192), (256, 8192), (256, 4)]),        # both k-contiguous, one tile, 16 slabs: <4,2,asm,2,2>
    ("xab,xbc,xcd->xad", [(3, 256, 4096), (3, 4096, 256), (3, 256, 8)]),   # a batch label
])
def test_large_tile_kernel_with_k_split_over_workgroups(einstr, shapes, monkeypatch):
    """A GEMM step whose 256 x 128 tiles cannot fill the chip while K is long (the root GEMMs of a sliced 2D grid with a
    rank's few slices in flight): K split over workgroups ON the large-tile LDS-DMA kernel, slabs added by the fixed-order
    reduce pass - against NumPy and against the same plan without the split (CTN_G_SPLITK=0), two replicas, twice for
    bit-identity."""
    rng = np.random.default_rng(17)
    ops = [(rng.standard_normal(s) / np.sqrt(s[-1])).astype(np.float32) for s in shapes]
    ref = np.einsum(einstr, *[o.astype(np.float64) for o in ops], optimize=["einsum_path", (0, 1), (0, 1)])
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("CTN_G_SPLITK", mode)
        E.clear_caches()
        bc = E.BatchedContraction(einstr, shapes, np.float32, optimize=((0, 1), (0, 1)), replicas=2)
        t, c = bc.run_host([ops, [2 * o for o in ops]])
        t2, c2 = bc.run_host([ops, [2 * o for o in ops]])
        assert np.array_equal(t, t2) and np.array_equal(c, c2)
        res[mode] = (t, c, bc.executor.step_tiles(), bc.plan.step_infos())
        bc.executor.close()
    monkeypatch.delenv("CTN_G_SPLITK")
    E.clear_caches()
    assert res["1"][2][0] == (256, 128) and res["1"][3][0]["tile_m"] == 256, (res["1"][2], res["1"][3][0])
    assert res["0"][2][0] != (256, 128)
    for mode in ("1", "0"):
        got = res[mode][0][0].astype(np.float64) * np.exp(float(res[mode][1][0]))
        assert np.max(np.abs(got - ref)) <= 2e-5 * np.max(np.abs(ref)), mode
        got2 = res[mode][0][1].astype(np.float64) * np.exp(float(res[mode][1][1]))
        assert np.max(np.abs(got2 - 8 * ref)) <= 2e-5 * np.max(np.abs(8 * ref)), mode


@pytest.mark.parametrize("einstr,shapes", [
    # (the operand popped from the higher position is the step's left one, reference einsum.py:344: the small matrix second)
    ("kn,km,n->m", [(256, 65536), (256, 256), (65536,)]),          # B row-contiguous along n: k_mfma_f32_ares<1>
    ("nk,mk,n->m", [(65536, 256), (256, 256), (65536,)]),          # B k-contiguous: <2>; A row-major
    ("kxy,km,xy->m", [(256, 2048, 48), (256, 256), (2048, 48)]),   # columns = two legs, the inner one 48 long: a tile's
                                                                    # 16-byte pieces are gathered through the table
    ("bkn,bkm,bn->bm", [(4, 256, 32768), (4, 256, 256), (4, 32768)]),   # four batch entries, each with its own A
])
def test_resident_left_operand_kernel(einstr, shapes, monkeypatch):
    """A 256 x 256 left operand against a very wide right one (the boundary absorptions of a 2D grid at bond 16): A lives
    in registers, a workgroup walks several 128-column tiles, only B streams (k_mfma_f32_ares).  Against NumPy and against
    the large-tile kernel (CTN_ARES=0), two replicas, twice for bit-identity."""
    rng = np.random.default_rng(29)
    ops = [(rng.standard_normal(s_) / 16.0).astype(np.float32) for s_ in shapes]
    ref = np.einsum(einstr, *[o.astype(np.float64) for o in ops], optimize=["einsum_path", (0, 1), (0, 1)])
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("CTN_ARES", mode)
        E.clear_caches()
        bc = E.BatchedContraction(einstr, shapes, np.float32, optimize=((0, 1), (0, 1)), replicas=2)
        t, c = bc.run_host([ops, [3 * o for o in ops]])
        t2, c2 = bc.run_host([ops, [3 * o for o in ops]])
        assert np.array_equal(t, t2) and np.array_equal(c, c2)
        res[mode] = (t, c, bc.executor.step_tiles())
        bc.executor.close()
    monkeypatch.delenv("CTN_ARES")
    E.clear_caches()
    assert res["1"][2][0][0] == 256 and res["1"][2][0][1] >= 512, res["1"][2]        # (256, 128 x tiles per workgroup)
    assert res["0"][2][0] in ((256, 128), (256, 256)), res["0"][2]
    for mode in ("1", "0"):
        got = res[mode][0][0].astype(np.float64) * np.exp(float(res[mode][1][0]))
        assert np.max(np.abs(got - ref)) <= 2e-5 * np.max(np.abs(ref)), mode
        got2 = res[mode][0][1].astype(np.float64) * np.exp(float(res[mode][1][1]))
        assert np.max(np.abs(got2 - 27 * ref)) <= 2e-5 * np.max(np.abs(27 * ref)), mode


def test_resident_left_operand_kernel_as_the_last_step(monkeypatch):
    """The same kernel on the step that writes the caller's result (no consumer behind it: k_finalize takes its collapsed
    partials), the result kept in caller order `mn`; CTN_ARES_NTW pins the tiles per workgroup."""
    rng = np.random.default_rng(31)
    einstr, shapes = "kn,km->mn", [(256, 131072), (256, 256)]
    ops = [(rng.standard_normal(s_) * 7.0).astype(np.float32) for s_ in shapes]
    ref = ops[1].astype(np.float64).T @ ops[0].astype(np.float64)
    res = {}
    for mode, ntw in (("1", "0"), ("1", "8"), ("0", "0")):
        monkeypatch.setenv("CTN_ARES", mode)
        monkeypatch.setenv("CTN_ARES_NTW", ntw)
        E.clear_caches()
        bc = E.BatchedContraction(einstr, shapes, np.float32, optimize=((0, 1),), replicas=1)
        t, c = bc.run_host([ops])
        res[mode + ntw] = (t, c, bc.executor.step_tiles())
        bc.executor.close()
    monkeypatch.delenv("CTN_ARES")
    monkeypatch.delenv("CTN_ARES_NTW")
    E.clear_caches()
    assert res["10"][2][0][0] == 256 and res["10"][2][0][1] >= 512, res["10"][2]
    assert res["18"][2][0] == (256, 1024), res["18"][2]
    assert res["00"][2][0] in ((256, 128), (256, 256)), res["00"][2]
    for key in res:
        got = res[key][0][0].astype(np.float64) * np.exp(float(res[key][1][0]))
        assert np.max(np.abs(got - ref)) <= 2e-5 * np.max(np.abs(ref)), key
    assert np.array_equal(res["10"][1], res["18"][1]) or abs(float(res["10"][1][0]) - float(res["18"][1][0])) < 1e-5


@pytest.mark.parametrize("mode", ["zipl", "zip64", "zip128"])
def test_zipper_forms_on_a_chain_with_uneven_bonds(mode, monkeypatch):
    """<phi|psi> where psi's bonds are 256, 272, 256, 256, 144 (phi's all 256): a pair is only taken by the fused forms when
    ITS shape fits (K1 = 256 for the latency form, |u| a multiple of 16 / 64 / 128), so fused and plain launches alternate
    along the chain: a latency-form pair whose successor is a plain step hands its slabs to k_zip_slab_sum, the next one
    starts from a plain tensor again, 17 and 9 u-blocks per network leave the XCD remap a remainder.  Against the oracle,
    two replicas, three runs for bit-identity."""
    from contractn_amd import TN
    from oracle import cpu_ref
    from tests import networks as nets

    n, phys = 7, 4
    psi_b = [256, 272, 256, 256, 144, 256]
    rng = np.random.default_rng(41)

    def cores(bonds):
        out = []
        for i in range(n):
            shape = (phys, bonds[0]) if i == 0 else (phys, bonds[-1]) if i == n - 1 else (phys, bonds[i - 1], bonds[i])
            out.append((rng.standard_normal(shape) / 16.0).astype(np.float32))
        return out

    tn = TN()
    a_nodes = nets.add_mps(tn, cores(psi_b))
    b_nodes = nets.add_mps(tn, cores([256] * (n - 1)))
    for x, y in zip(a_nodes, b_nodes):
        tn.connect_nodes(x, y, 0, 0)
    from contractn_amd.paths import ssa_to_linear

    path = ssa_to_linear(nets.zipper_path(n), 2 * n)
    shapes = [p.shape for p in tn.params]
    sets = [list(tn.params), [(p * np.float32(1.5)).astype(np.float32) for p in tn.params]]
    env = {"zipl": {"CTN_ZIP": "0", "CTN_ZIPL": "1"}, "zip64": {"CTN_ZIP": "2", "CTN_ZIPL": "0"}, "zip128": {"CTN_ZIP": "1", "CTN_ZIPL": "0"}}[mode]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    E.clear_caches()
    bc = E.BatchedContraction(tn.einsum_str, shapes, np.float32, optimize=path, replicas=2)
    t, c = bc.run_host(sets)
    for _ in range(2):
        t2, c2 = bc.run_host(sets)
        assert np.array_equal(t, t2) and np.array_equal(c, c2)
    tiles = bc.executor.step_tiles()
    bc.executor.close()
    for k in env:
        monkeypatch.delenv(k)
    E.clear_caches()
    fused = [tl for tl in tiles if tl in ((32, 256), (64, 256), (512, 128), (512, 256))]
    assert len(fused) >= (2 if mode == "zipl" else 1), tiles
    for r in range(2):
        rt, rc = cpu_ref.contract(tn.einsum_str, *sets[r], path=path, split_format=True)
        assert float(t[r]) == float(rt) and abs(float(c[r]) - float(rc)) <= 1e-4, (mode, r, float(c[r]), float(rc))


# ---- a batched MPS as ONE launch (k_sweep_f32): 16 inputs per workgroup walk every site of the chain --------------------
@pytest.mark.parametrize("sites,batch,replicas,spread,bond,phys", [
    (6, 64, 1, 1.0, 256, 4), (9, 160, 2, 1.0, 256, 4), (7, 80, 1, 1e6, 256, 4),
    (7, 144, 1, 1.0, 256, 2),        # physical dimension 2
    (8, 272, 2, 1e4, 128, 4),        # bond 128: 2 ranges of r x 4 parts of l
    (9, 528, 1, 1.0, 64, 4),         # bond 64: a 4-wave workgroup, one group of l per wave
    (12, 1040, 1, 1.0, 64, 2),
    (5, 64, 1, 1.0, 512, 4),         # bond 512: 8 ranges of r, no hand-over
    (6, 88, 1, 1.0, 512, 2),         # ... 88 inputs: the last block holds 8
    (7, 264, 1, 1.0, 128, 2),        # 264 inputs: the last block holds 8
    (6, 100, 2, 1e3, 256, 4),        # 100 inputs: the last block holds 4
])
def test_sweep_of_a_batched_mps_matches_the_per_site_launches(sites, batch, replicas, spread, bond, phys, monkeypatch):
    """The reference's ML workload (README Fig. 1d; BASELINE config 3b) at bonds 64 ... 512, physical dimension 2 and 4,
    batches that are not multiples of 16: with
    CTN_SWEEP=1 the interior sites - one epilogue-summed GEMM step each - run as ONE k_sweep_f32<bond, phys> launch in which
    every block of 16 inputs rescales by its OWN mean; k_sweep_z / k_sweep_finish reconstruct the reference's
    per-step rescale factors (mean over ALL inputs, einsum.py:97-106).  Against the per-site launches (CTN_SWEEP=0):
    result, log-scale and EVERY step's rescale factor; against the oracle; replicas; blocks of very different
    magnitude (`spread`: the inputs of the second half of the batch are that much larger)."""
    from contractn_amd import TN
    from contractn_amd.paths import ssa_to_linear
    from oracle import cpu_ref
    from tests import networks as nets

    tn, inputs = nets.batched_mps(TN, sites, bond, phys, batch, dtype=np.float32, seed=4)
    path = ssa_to_linear(nets.batched_mps_path(sites), 2 * sites)
    ops0 = [np.asarray(o) for o in E.make_arg_packer(tn)(tn.params, inputs)]
    shapes = [o.shape for o in ops0]
    rng = np.random.default_rng(7)
    sets = []
    for r in range(replicas):
        ops = [(rng.standard_normal(sh) * (0.25 if len(sh) == 2 and sh[0] == batch else 1.0 / np.sqrt(bond))).astype(np.float32) for sh in shapes]
        if spread != 1.0:
            for o in ops:
                if o.shape == (batch, phys):
                    o[batch // 2:] *= np.float32(spread ** (1.0 / sites))
        sets.append(ops)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("CTN_SWEEP", mode)
        E.clear_caches()
        bc = E.BatchedContraction(tn.einsum_str, shapes, np.float32, optimize=path, replicas=replicas)
        t, c = bc.run_host(sets)
        t2, c2 = bc.run_host(sets)                     # graph capture / replay: the same bits
        assert np.array_equal(t, t2) and np.array_equal(c, c2)
        res[mode] = (t, c, bc.executor.step_tiles(), bc.executor.fetch()[1])
        bc.executor.close()
    tiles = res["1"][2]
    whole = [s for s, tl in enumerate(tiles) if tl == (16, bond * phys)]
    assert len(whole) == 1 and sum(tl == (1, 1) for tl in tiles) >= sites - 3, tiles
    assert not any(tl == (16, bond * phys) for tl in res["0"][2])
    for r in range(replicas):
        r0, r1 = np.asarray(res["0"][3][r]), np.asarray(res["1"][3][r])
        assert np.array_equal(r0 == 0.0, r1 == 0.0)                               # the same steps are rescaled
        nz = r0 != 0.0
        assert np.max(np.abs(r1[nz] / r0[nz] - 1.0)) <= 2e-5, (r0, r1)            # ... by the same factors
        assert abs(float(res["0"][1][r]) - float(res["1"][1][r])) <= 1e-4
        a = res["0"][0][r].astype(np.float64) * np.exp(float(res["0"][1][r]))
        b = res["1"][0][r].astype(np.float64) * np.exp(float(res["1"][1][r]))
        assert np.max(np.abs(a - b)) <= 2e-5 * np.max(np.abs(a))
        rt, rc = cpu_ref.contract(tn.einsum_str, *sets[r], path=path, split_format=True)
        ref = np.asarray(rt, dtype=np.float64) * np.exp(float(rc))
        assert np.max(np.abs(b - ref)) <= 1e-4 * np.max(np.abs(ref))
        assert abs(float(np.mean(np.abs(res["1"][0][r]))) - 1.0) < 1e-5 and abs(float(res["1"][1][r]) - float(rc)) <= 1e-4
    monkeypatch.delenv("CTN_SWEEP")
    E.clear_caches()


def test_sweep_steps_aside_in_eager_mode_and_survives_an_eager_repeat(monkeypatch):
    """An executor that walks a batched MPS as one k_sweep_f32 launch: (1) forced into the eager rescale mode it takes
    the per-site launches again (the sweep keeps its own products in range, but eager mode means the reference's literal
    order for EVERY step) and gives the lazy run's numbers to rounding; (2) every operand 1e9 times larger: whether the
    fetch has to repeat the contraction eagerly or not, the value is the oracle's; (3) the next, tame operands run
    lazily - with the sweep - again, to the same bits as before."""
    from contractn_amd import TN
    from contractn_amd.paths import ssa_to_linear
    from oracle import cpu_ref
    from tests import networks as nets

    sites, batch, bond, phys = 7, 64, 256, 4
    monkeypatch.setenv("CTN_SWEEP", "1")
    E.clear_caches()
    tn, inputs = nets.batched_mps(TN, sites, bond, phys, batch, dtype=np.float32, seed=4)
    path = ssa_to_linear(nets.batched_mps_path(sites), 2 * sites)
    shapes = [np.asarray(o).shape for o in E.make_arg_packer(tn)(tn.params, inputs)]
    rng = np.random.default_rng(11)
    tame = [(rng.standard_normal(sh) * (0.25 if sh == (batch, phys) else 1.0 / 16.0)).astype(np.float32) for sh in shapes]

    def value(t, c):
        return np.asarray(t, dtype=np.float64) * np.exp(float(c))

    lazy = E.BatchedContraction(tn.einsum_str, shapes, np.float32, optimize=path, replicas=1)
    t_l, c_l = lazy.run_host([tame])
    assert (16, 1024) in lazy.executor.step_tiles() and lazy.executor.eager_reruns() == 0
    eager = E.BatchedContraction(tn.einsum_str, shapes, np.float32, optimize=path, replicas=1)
    eager.executor.set_rescale_mode(1)
    t_e, c_e = eager.run_host([tame])
    assert (16, 1024) not in eager.executor.step_tiles() and eager.executor.eager_reruns() == 0
    a, b = value(t_l[0], c_l[0]), value(t_e[0], c_e[0])
    assert np.max(np.abs(a - b)) <= 2e-5 * np.max(np.abs(b))
    # (2) every tensor 1e9 times larger (the reference, like the eager mode, normalises after every step)
    huge = [(o * np.float32(1e9)).astype(np.float32) for o in tame]
    t_h, c_h = lazy.run_host([huge])
    rt, rc = cpu_ref.contract(tn.einsum_str, *huge, path=path, split_format=True)
    assert np.all(np.isfinite(t_h)) and abs(float(c_h[0]) - float(rc)) <= 1e-3
    assert np.max(np.abs(t_h[0] - np.asarray(rt))) <= 1e-3 * np.max(np.abs(np.asarray(rt)))
    # (3) tame operands again: the same bits as the first lazy run, sweep and all
    reruns = lazy.executor.eager_reruns()
    t_2, c_2 = lazy.run_host([tame])
    assert np.array_equal(t_2, t_l) and np.array_equal(c_2, c_l) and lazy.executor.eager_reruns() == reruns
    assert (16, 1024) in lazy.executor.step_tiles()
    lazy.executor.close(); eager.executor.close()
    monkeypatch.delenv("CTN_SWEEP")
    E.clear_caches()


@pytest.mark.parametrize("batch", [256, 1000])
def test_sweep_is_not_taken_across_interleaved_chains(batch, monkeypatch):
    """Two batched MPS hanging on one batch hyperedge, walked ALTERNATELY (a site of chain A, a site of chain B, ...):
    between two members of either chain lies a launched step of the other, whose result the arena may place where the
    chain's first input was (released right after its first member).  A sweep launches at its last member's position and
    reads that input there - so such a run must not be taken (round-3 advice, engine.hip sweep_match).  With CTN_SWEEP=1
    the value must be the per-site launches' and the oracle's, and no step may report the sweep's tile."""
    from contractn_amd.paths import ssa_to_linear
    from contractn_amd.utils import get_symbol
    from oracle import cpu_ref

    n, bond, phys = 6, 128, 4
    sym = iter(get_symbol(i) for i in range(200))
    b = next(sym)
    terms, shapes = [None] * (4 * n), [None] * (4 * n)
    for chain in range(2):
        base = 2 * n * chain
        left = None
        for i in range(n):
            p_, right = next(sym), (next(sym) if i + 1 < n else None)
            legs = [p_] + ([left] if left else []) + ([right] if right else [])
            terms[base + i] = "".join(legs)
            shapes[base + i] = (phys,) + (bond,) * (len(legs) - 1)
            terms[base + n + i] = b + p_
            shapes[base + n + i] = (batch, phys)
            left = right
    einstr = ",".join(terms) + "->" + b
    ssa, nxt = [], 4 * n
    cur = [None, None]
    for chain in range(2):
        ssa.append((2 * n * chain, 2 * n * chain + n))
        cur[chain] = nxt
        nxt += 1
    for i in range(1, n):
        for chain in range(2):
            base = 2 * n * chain
            ssa.append((cur[chain], base + i))
            ssa.append((nxt, base + n + i))
            cur[chain] = nxt + 1
            nxt += 2
    ssa.append((cur[0], cur[1]))
    path = ssa_to_linear(ssa, 4 * n)
    rng = np.random.default_rng(31)
    ops = [(rng.standard_normal(sh) * (0.25 if sh == (batch, phys) else 1.0 / np.sqrt(bond))).astype(np.float32) for sh in shapes]
    rt, rc = cpu_ref.contract(einstr, *ops, path=path, split_format=True)
    ref = np.asarray(rt, dtype=np.float64) * np.exp(float(rc))
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("CTN_SWEEP", mode)
        E.clear_caches()
        bc = E.BatchedContraction(einstr, shapes, np.float32, optimize=path, replicas=1)
        for _ in range(3):                                   # eager launches, graph capture, replay
            t, c = bc.run_host([ops])
        res[mode] = np.asarray(t[0], dtype=np.float64) * np.exp(float(c[0]))
        assert not any(tl[0] == 16 and tl[1] == bond * phys for tl in bc.executor.step_tiles())
        bc.executor.close()
    monkeypatch.delenv("CTN_SWEEP")
    E.clear_caches()
    assert np.max(np.abs(res["1"] - ref)) <= 1e-4 * np.max(np.abs(ref))
    assert np.array_equal(res["0"], res["1"])


# ---- a full dot of a tensor with a transposed one (k_dot_tr) -----------------------------------------------------------------
@pytest.mark.parametrize("einstr,shapes,dtype", [
    ("ab,ba->", [(512, 512), (512, 512)], np.float32),           # the closing dot of a sliced 8 x 8 PEPS (D = 8)
    ("ab,ba->", [(256, 1024), (1024, 256)], np.float32),
    ("ab,ba->", [(1024, 64), (64, 1024)], np.float64),
    ("xab,xba->x", [(3, 256, 256), (3, 256, 256)], np.float32),  # a batch (hyperedge) label: three outputs
    ("abc,cab->", [(64, 32, 32), (32, 64, 32)], np.float32),     # k = (a, (b, c)) against ((c), (a, b)): NOT a plain transpose
])
def test_full_dot_against_a_transposed_tensor(einstr, shapes, dtype, monkeypatch):
    """Both operands are read along their own unit-stride index and one tile turns round in LDS; against NumPy and against
    the 4-byte gathers of k_dot_split (CTN_DOT_TR=0), two replicas, twice for bit-identity."""
    rng = np.random.default_rng(23)
    ops = [rng.standard_normal(s).astype(dtype) for s in shapes]
    ref = np.einsum(einstr, *[o.astype(np.float64) for o in ops])
    tol = 2e-4 if dtype == np.float32 else 1e-11
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("CTN_DOT_TR", mode)
        E.clear_caches()
        bc = E.BatchedContraction(einstr, shapes, dtype, optimize=((0, 1),), replicas=2)
        t, c = bc.run_host([ops, [2 * o for o in ops]])
        t2, c2 = bc.run_host([ops, [2 * o for o in ops]])
        assert np.array_equal(t, t2) and np.array_equal(c, c2)
        for r, f in ((0, 1.0), (1, 4.0)):
            got = np.asarray(t[r], dtype=np.float64) * np.exp(float(c[r]))
            assert np.max(np.abs(got - f * ref)) <= tol * f * max(1.0, np.max(np.abs(ref)) * 10), (mode, r)
        res[mode] = np.asarray(t[0], dtype=np.float64) * np.exp(float(c[0]))
        bc.executor.close()
    assert np.max(np.abs(res["0"] - res["1"])) <= tol * max(1.0, np.max(np.abs(ref)) * 10)
    monkeypatch.delenv("CTN_DOT_TR")
    E.clear_caches()


@pytest.mark.parametrize("seed", range(12))
def test_wide_absorptions_in_random_layouts(seed):
    """A boundary tensor absorbing a site in every arrangement a 2D grid produces: the wide operand's legs in random order
    (the contracted ones anywhere among them), the site tensor's four legs in random order, either operand first.  K = 256
    and 256 free values on the small side, so the planner puts the small operand left and the step runs on
    k_mfma_f32_ares when the wide operand is unit-stride along a free or a contracted leg, on the general kernels when
    it is not; then one more absorption on the result (its layout is the engine's).  Against torch.einsum in float64."""
    import torch

    rng = np.random.default_rng(1000 + seed)
    legs = "abcdefg"                                   # wide tensor: 4 free legs (16, 16, 16, 8..32) + 2 contracted (16, 16)
    free_ext = [16, 16, 16, int(rng.choice([8, 16, 32]))]
    wide_legs = list("abcd") + list("kl")
    ext = dict(zip("abcd", free_ext), k=16, l=16, m=16, n=16, p=16, q=16)
    order = rng.permutation(len(wide_legs))
    wide = "".join(wide_legs[i] for i in order)
    site1 = "".join(rng.permutation(list("klmn")))      # contracts k, l; new legs m, n
    # second absorption: contracts one old free leg and one new leg of the first result
    site2 = "".join(rng.permutation(list("anpq")))
    out = "".join(rng.permutation(list("bcdmpq")))
    terms = [wide, site1, site2]
    if rng.integers(2):
        terms = [site1, wide, site2]
    einstr = ",".join(terms) + "->" + out
    shapes = [tuple(ext[c] for c in t) for t in terms]
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    ops = [torch.randn(s, generator=gen, device="cuda") * 0.25 for s in shapes]
    t, c = contract(einstr, *ops, optimize=((0, 1), (0, 1)), split_format=True)
    ref = torch.einsum(einstr, *[o.double() for o in ops])
    got = t.double() * torch.exp(c.double())
    assert float((got - ref).abs().max() / ref.abs().max()) <= 2e-5, einstr
    E.clear_caches()
