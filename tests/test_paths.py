"""Path finder + contraction-list format (restatement of opt_einsum.contract_path(einsum_call=True))."""
import numpy as np
import pytest

from contractn_amd import paths
from contractn_amd.einsum import lower_contraction_list
from oracle import cpu_ref
from tests.helpers import load_golden


def run_list(einstr, ops, clist):
    out, c, _ = cpu_ref.core_contract(ops, clist)
    return out * np.exp(c)


@pytest.mark.parametrize("optimize", ["auto", "greedy", "optimal"])
@pytest.mark.parametrize("name", ["cp_r5_f64", "tucker_r5_f64", "mps_open_random_f64", "batched_mps_f64",
                                  "peps3x3_D2_f64"])
def test_found_paths_are_valid(name, optimize):
    g = load_golden(name)
    ops = g["operands"]
    if optimize == "optimal" and len(ops) > 8:
        pytest.skip("exhaustive search is for small networks")
    clist = paths.contraction_list(g["einsum_str"], [o.shape for o in ops], optimize=optimize)
    assert len(clist) == len(ops) - 1
    np.testing.assert_allclose(run_list(g["einsum_str"], ops, clist), g["plain"], rtol=1e-10)


def test_contraction_list_format_matches_oracle_builder():
    g = load_golden("mps_overlap_6x8x3_f64")
    shapes = [o.shape for o in g["operands"]]
    mine = paths.contraction_list(g["einsum_str"], shapes, optimize=g["path"])
    ref = cpu_ref.contraction_list(g["einsum_str"], shapes, g["path"])
    for a, b in zip(mine, ref):
        assert a[0] == b[0] and set(a[1]) == set(b[1]) and a[2] == b[2] and a[4] == b[4]
    # popped order: left = higher position; positions sorted descending
    assert all(list(c[0]) == sorted(c[0], reverse=True) for c in mine)


def test_blas_flag_routing():
    cl = paths.contraction_list("ab,bc->ac", [(2, 3), (3, 4)], optimize=((0, 1),))
    assert cl[0][4] == "TDOT"
    cl = paths.contraction_list("a,a->a", [(2,), (2,)], optimize=((0, 1),))
    assert cl[0][4] is False  # shared label kept: hyperedge -> einsum route
    cl = paths.contraction_list("aa->", [(3, 3)], optimize=((0,),))
    assert cl[0][4] is False


def test_intermediate_label_order_is_dim_then_symbol():
    cl = paths.contraction_list("ab,bc,cd->ad", [(5, 2), (2, 3), (3, 4)], optimize=((0, 1), (0, 1)))
    assert cl[0][2] == "bc,ab->ca"   # sizes c=3 < a=5
    assert cl[1][2].endswith("->ad")  # final step uses the caller's order


def test_ssa_lowering():
    cl = paths.contraction_list("ab,bc,cd->ad", [(5, 2), (2, 3), (3, 4)], optimize=((0, 1), (0, 1)))
    in_labels, steps = lower_contraction_list(3, cl)
    assert in_labels == [(97, 98), (98, 99), (99, 100)]
    assert steps[0][:2] == (1, 0) and steps[1][:2] == (3, 2)
    assert paths.ssa_to_linear([(0, 1), (3, 2)], 3) == ((0, 1), (0, 1))


def test_errors():
    with pytest.raises(ValueError):
        paths.contraction_list("ab,bc->ac", [(2, 3), (4, 2)])
    with pytest.raises(ValueError):
        paths.contraction_list("ab,bc->az", [(2, 3), (3, 2)])
    with pytest.raises(KeyError):
        paths.contraction_list("ab,bc,cd->ad", [(2, 3), (3, 2), (2, 2)], optimize="nonsense")


def test_greedy_scales_to_1000_operands():
    g = load_golden("readme_chain1000")
    cl = paths.contraction_list(g["einsum_str"], [o.shape for o in g["operands"]], optimize="greedy")
    assert len(cl) == 1000
