"""C-ABI library: loads, exports every symbol of include/ctn_abi.h, plans build and validate on the host."""
import ctypes
import os
import re

import numpy as np
import pytest

from contractn_amd import einsum as E
from contractn_amd import engine
from tests.helpers import ROOT, load_golden


def header_symbols():
    text = open(os.path.join(ROOT, "include", "ctn_abi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ctn_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = engine.load_library()
    declared = header_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in ctn_abi.h but not exported"
    assert sorted(engine.ABI_SYMBOLS) == declared
    assert lib.ctn_version() == 5


def test_library_has_no_torch_dependency():
    import subprocess

    out = subprocess.run(["ldd", engine.LIB_PATH], capture_output=True, text=True).stdout
    assert "torch" not in out and "libamdhip64" in out


def _plan(einstr, shapes, path, dtype="float32"):
    clist = E._contract_path(einstr, tuple(shapes), optimize=path, memory_limit=None, use_blas=True)
    return E._native_plan(clist, tuple(shapes), dtype)


def test_plan_for_headline_network_lowers_to_mfma_gemms():
    """Config 3a shape (reduced site count): every bulk step is an MFMA GEMM with vector loads."""
    from contractn_amd import TN
    from contractn_amd.paths import ssa_to_linear
    from tests import networks as nets

    tn, ssa = nets.mps_overlap(TN, 5, 256, 4, dtype=np.float32)
    shapes = [p.shape for p in tn.params]
    plan = _plan(tn.einsum_str, shapes, ssa_to_linear(ssa, 10))
    infos = plan.step_infos()
    assert plan.n_steps == 9 and plan.out_shape == ()
    mfma = [i for i in infos if i["kernel"] == 2]
    assert len(mfma) == 7                 # six bulk GEMMs and the closing 256 x 4 x 256 step (a masked tile, not a streaming step)
    assert {(i["m"], i["n"], i["k"]) for i in mfma} == {(256, 1024, 256), (256, 256, 1024), (256, 4, 256)}
    assert all(i["mode_a"] in (1, 2) and i["mode_b"] in (1, 2) for i in mfma)
    assert all(i["partials"] <= 64 for i in infos)
    # flop count: 2*M*N*K + 3*numel(out) per step
    assert mfma[0]["flops"] == 2.0 * mfma[0]["m"] * mfma[0]["n"] * mfma[0]["k"] + 3.0 * mfma[0]["out_numel"]
    assert plan.flops == pytest.approx(sum(i["flops"] for i in infos))
    assert plan.bytes_min == sum(int(np.prod(s)) for s in shapes) * 4 + 4
    assert plan.workspace_bytes(8) > 8 * (256 * 1024 * 4)


def test_headline_flop_count_matches_survey():
    """100 sites, D=256, d=4: 98*4*d*D^3 + 2*2dD^2 + 2dD = 26,307,725,312 multiply-adds*2."""
    from contractn_amd import TN
    from contractn_amd.paths import ssa_to_linear
    from tests import networks as nets

    tn, ssa = nets.mps_overlap(TN, 100, 8, 4, dtype=np.float32)  # small bond: same structure
    plan = _plan(tn.einsum_str, [p.shape for p in tn.params], ssa_to_linear(ssa, 200))
    d, D = 4, 8
    gemm = 98 * 4 * d * D ** 3 + 2 * 2 * d * D ** 2 + 2 * d * D
    stab = 3 * sum(i["out_numel"] for i in plan.step_infos())
    assert plan.n_steps == 199
    assert plan.flops == gemm + stab


def test_hyperedge_steps_have_batch_labels_and_no_delta_tensor():
    g = load_golden("readme_copy101")
    plan = _plan(g["einsum_str"], [o.shape for o in g["operands"]], g["path"], "float64")
    infos = plan.step_infos()
    assert len(infos) == 99
    assert all(i["batch"] == 2 and i["m"] == i["n"] == i["k"] == 1 and i["kernel"] == 0 for i in infos)
    assert plan.bytes_min == 100 * 2 * 8 + 2 * 8  # only the vectors and the output: no identity tensor


def test_trace_sumout_and_unary_steps_lower():
    plan = _plan("aa->", [(3, 3)], ((0,),), "float64")
    i = plan.step_info(0)
    assert (i["batch"], i["m"], i["n"], i["k"]) == (1, 1, 1, 3)
    plan = _plan("ab->ab", [(2, 3)], ((0,),), "float64")
    assert plan.out_shape == (2, 3)
    plan = _plan("abc,cd->abd", [(4, 3, 5), (5, 2)], ((0, 1),), "float64")
    assert plan.out_shape == (4, 3, 2)


@pytest.mark.parametrize("bad,exc", [
    (dict(in_dims=[(2, 3), (4, 2)]), ValueError),            # label extent mismatch
    (dict(steps=[(0, 1, (ord("z"),))]), AssertionError),     # output label in neither operand
    (dict(steps=[(0, 0, (ord("a"),))]), AssertionError),     # operand used twice
    (dict(steps=[(0, 5, (ord("a"),))]), AssertionError),     # id out of range
])
def test_plan_validation_errors(bad, exc):
    args = dict(in_labels=[(ord("a"), ord("b")), (ord("b"), ord("c"))], in_dims=[(2, 3), (3, 2)],
                steps=[(0, 1, (ord("a"), ord("c")))])
    args.update(bad)
    with pytest.raises(exc):
        engine.Plan("float32", args["in_labels"], args["in_dims"], args["steps"])


def test_unsupported_dtype_rejected():
    with pytest.raises(TypeError):
        engine.Plan("complex64", [(1,)], [(2,)], [(0, -1, (1,))])


def test_no_cpu_fallback_without_device():
    if engine.device_count() > 0:
        pytest.skip("a GPU is present")
    plan = _plan("ab,bc->ac", [(2, 3), (3, 2)], ((0, 1),))
    with pytest.raises(RuntimeError, match="no HIP device"):
        engine.Executor(plan)
    from contractn_amd import contract

    with pytest.raises(RuntimeError):
        contract("ab,bc->ac", np.ones((2, 3), np.float32), np.ones((3, 2), np.float32))


def test_raw_ctypes_error_reporting():
    lib = engine.load_library()
    handle = ctypes.c_void_p()
    rc = lib.ctn_plan_create(None, ctypes.byref(handle))
    assert rc == -1 and b"NULL" in lib.ctn_last_error()


def test_planner_selects_large_tile_kernels():
    """ctn_step_info.tile_m / tile_n: the planner's choice of the LDS-DMA large-tile kernels (host-only)."""
    from contractn_amd import einsum as E

    def info(einstr, shapes, dtype):
        clist = E._contract_path(einstr, tuple(shapes), optimize=((0, 1),), memory_limit=None, use_blas=True)
        return E._native_plan(clist, tuple(shapes), dtype).step_infos()[0]

    # both operands unit-stride along their free index, full tiles: fp32 256 x 128, fp64 128 x 128
    i = info("km,kn->mn", [(256, 256), (256, 1024)], "float32")
    assert (i["kernel"], i["tile_m"], i["tile_n"]) == (2, 256, 128)
    i = info("km,kn->mn", [(256, 256), (256, 1024)], "float64")
    assert (i["kernel"], i["tile_m"], i["tile_n"]) == (3, 128, 128)
    # ragged M / N / K are fine as long as 256-row tiles do not pad much more than 128-row ones
    assert info("km,kn->mn", [(100, 400), (100, 200)], "float32")["tile_m"] == 256
    assert info("km,kn->mn", [(100, 300), (100, 128)], "float32")["tile_m"] == 64       # 300 -> 512 rows: too much padding; 5 x 64 rows pad least
    assert info("km,kn->mn", [(100, 384), (100, 128)], "float32")["tile_m"] == 128
    # skinny rows against a long N (PEPS boundary absorption): 64-row tiles, no half-masked MFMAs
    i = info("mk,kn->mn", [(64, 64), (64, 32768)], "float32")
    assert (i["kernel"], i["tile_m"], i["tile_n"], i["blocks"]) == (2, 64, 128, 256)
    # a k-contiguous (row-major) operand is fine too, with any K
    assert info("mk,kn->mn", [(256, 256), (256, 1024)], "float32")["tile_m"] == 256
    assert info("mk,kn->mn", [(256, 40), (40, 1024)], "float32")["tile_m"] == 256
    assert info("mk,kn->mn", [(256, 256), (256, 1024)], "float64")["tile_n"] == 128
    assert info("mk,kn->mn", [(256, 42), (42, 1024)], "float64")["tile_n"] == 128
    # fewer than two k-tiles
    assert info("km,kn->mn", [(16, 256), (16, 256)], "float32")["tile_m"] == 128
    assert info("km,kn->mn", [(8, 128), (8, 128)], "float64")["tile_n"] == 64


def test_planner_puts_a_small_operand_left_of_a_very_wide_one():
    """K = 256, 256 free values on one side and >= 32768 on the other (a 2D grid's boundary absorbing a site): the small
    operand becomes the step's left one whatever the caller's order - the engine keeps it in registers (k_mfma_f32_ares)."""
    from contractn_amd import einsum as E

    def first(einstr, shapes):
        clist = E._contract_path(einstr, tuple(shapes), optimize=((0, 1), (0, 1)), memory_limit=None, use_blas=True)
        return E._native_plan(clist, tuple(shapes), "float32").step_infos()[0]

    a = first("km,kn,n->m", [(256, 256), (256, 65536), (65536,)])      # the wide operand popped second: it would be the left one
    b = first("kn,km,n->m", [(256, 65536), (256, 256), (65536,)])
    for i in (a, b):
        assert (i["m"], i["n"], i["k"], i["tile_m"]) == (256, 65536, 256, 256), i
    assert a["swapped"] == 1 and b["swapped"] == 0
    c = first("km,kn,n->m", [(256, 128), (256, 65536), (65536,)])      # 128 free values: not this rule
    assert c["m"] == 65536 and c["swapped"] == 0


def test_results_of_2_to_32_elements_keep_their_small_group_whole():
    """256 rows against 2^24 columns (three legs of 256): the outermost column leg becomes a batch label, outermost in the
    result, and every batch entry is a 256 x 65536 matrix - not 2^24 columns against ONE row per batch entry."""
    from contractn_amd import einsum as E

    shapes = [(256, 256), (256, 256, 256, 256), (256,)]
    clist = E._contract_path("km,xkyz,z->mxy", tuple(shapes), optimize=((0, 1), (0, 1)), memory_limit=None, use_blas=True)
    i = E._native_plan(clist, tuple(shapes), "float32").step_infos()[0]
    assert (i["batch"], i["m"], i["n"], i["k"]) == (256, 256, 65536, 256), i
    assert i["tile_m"] == 256 and i["out_numel"] == 2 ** 32
