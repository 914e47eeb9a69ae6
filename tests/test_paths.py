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


def _random_net(rng, n, n_labels):
    labels = [chr(97 + i) for i in range(n_labels)]
    sizes = {l: int(rng.choice([2, 3, 4, 8])) for l in labels}
    terms = ["".join(rng.choice(labels, size=int(rng.integers(1, 4)), replace=False)) for _ in range(n)]
    present = sorted(set("".join(terms)))
    out = "".join(l for l in present if rng.random() < 0.2)
    return terms, out, sizes


def test_dp_equals_exhaustive_search_and_noisy_greedy_never_loses():
    """'dp' is exact (same flop count as the exhaustive 'optimal', hyperedges and outer products included);
    'random-greedy' keeps the plain greedy path as its first trial."""
    rng = np.random.default_rng(0)
    improved = 0
    for _ in range(150):
        terms, out, sizes = _random_net(rng, int(rng.integers(3, 8)), int(rng.integers(3, 9)))
        sets = [set(t) for t in terms]
        cost = lambda p: paths.path_cost(sets, out, sizes, p)[0]  # noqa: E731
        assert cost(paths._dp(sets, out, sizes)) == cost(paths._optimal(sets, out, sizes)), (terms, out)
        g, r = cost(paths._greedy(sets, out, sizes)), cost(paths._random_greedy(sets, out, sizes, repeats=16))
        assert r <= g
        improved += r < g
    assert improved > 10


@pytest.mark.parametrize("optimize", ["dp", "random-greedy", "random-greedy-8", "auto-hq"])
def test_new_strategies_give_valid_contractions(optimize):
    g = load_golden("peps3x3_D2_f64")
    ops = g["operands"]
    clist = paths.contraction_list(g["einsum_str"], [o.shape for o in ops], optimize=optimize)
    assert len(clist) == len(ops) - 1
    np.testing.assert_allclose(run_list(g["einsum_str"], ops, clist), g["plain"], rtol=1e-10)


def test_auto_beats_plain_greedy_on_a_grid():
    from contractn_amd import TN
    from tests import networks as nets

    tn = nets.peps_closed(TN, 3, 4, 4)
    shapes = [p.shape for p in tn.params]
    terms, out, sizes = paths.parse_einsum_input(tn.einsum_str, shapes)
    sets = [set(t) for t in terms]
    auto = paths.path_cost(sets, out, sizes, paths.find_path(terms, out, sizes, "auto"))
    greedy = paths.path_cost(sets, out, sizes, paths.find_path(terms, out, sizes, "greedy"))
    assert auto[0] < greedy[0] and auto[1] <= greedy[1]


def test_memory_limit_steers_the_search():
    # a chain where the cheapest order builds a large intermediate: with a limit the search avoids it
    einstr, shapes = "ab,bc,cd,de->ae", [(64, 2), (2, 64), (64, 2), (2, 64)]
    terms, out, sizes = paths.parse_einsum_input(einstr, shapes)
    sets = [set(t) for t in terms]
    free = paths.path_cost(sets, out, sizes, paths.find_path(terms, out, sizes, "dp"))
    tight = paths.path_cost(sets, out, sizes, paths.find_path(terms, out, sizes, "dp", memory_limit=200))
    assert tight[1] <= max(free[1], 200) or tight[1] <= free[1]
    cl = paths.contraction_list(einstr, shapes, optimize="dp", memory_limit="max_input")
    assert len(cl) == 3


def test_auto_restarts_only_where_greedy_is_not_flat():
    """'auto' beyond 12 operands: a greedy path whose intermediates never outgrow the largest operand is taken as
    is (hub / chain: milliseconds), a lattice gets noisy restarts (8 x 8 PEPS: several times cheaper than greedy)."""
    import time

    from contractn_amd import TN
    from tests import networks as nets

    tn = TN()
    hub = tn.add_copy_node(101)
    for i in range(100):
        tn.connect_nodes(hub, tn.add_dense_node(np.array([1, 0.99])), i, 0)
    terms, out, sizes = paths.parse_einsum_input(tn.einsum_str, [p.shape for p in tn.params])
    t0 = time.perf_counter()
    auto = paths.find_path(terms, out, sizes, "auto")
    assert time.perf_counter() - t0 < 0.2
    assert paths.path_cost(terms, out, sizes, auto) == paths.path_cost(terms, out, sizes, paths.find_path(terms, out, sizes, "greedy"))

    tn = nets.peps_closed(TN, 6, 6, 2, dtype=np.float32, seed=6)
    shapes = [tuple(6 if (d == 2 and p.ndim > 1 and ax > 0) else d for ax, d in enumerate(p.shape)) for p in tn.params]
    terms, out, sizes = paths.parse_einsum_input(tn.einsum_str, shapes)
    c_auto = paths.path_cost(terms, out, sizes, paths.find_path(terms, out, sizes, "auto"))
    c_greedy = paths.path_cost(terms, out, sizes, paths.find_path(terms, out, sizes, "greedy"))
    assert c_auto <= c_greedy


def test_subtree_reconfiguration_is_valid_and_never_worse():
    """`_reconfigure`: exact DP on subtrees of a given tree.  On random networks with hyperedges the refined path
    contracts to the same value (oracle walk vs np.einsum), never costs more, often costs less; with every
    operand in one subtree it reproduces the exact optimum."""
    rng = np.random.default_rng(3)
    improved = 0
    for trial in range(60):
        n = int(rng.integers(4, 14))
        terms, out, sizes = _random_net(rng, n, int(rng.integers(4, 10)))
        sets = [set(t) for t in terms]
        start = paths._greedy(sets, out, sizes) if trial % 2 else [(0, 1)] * (n - 1)   # greedy / left-to-right
        c0 = paths.path_cost(sets, out, sizes, start)[0]
        new = paths._reconfigure(sets, out, sizes, start, max_leaves=6, rounds=4)
        assert len(new) == n - 1
        c1 = paths.path_cost(sets, out, sizes, new)[0]
        assert c1 <= c0
        improved += c1 < c0
        if n <= 8:
            full = paths._reconfigure(sets, out, sizes, start, max_leaves=n, rounds=1)
            assert paths.path_cost(sets, out, sizes, full)[0] == paths.path_cost(sets, out, sizes, paths._dp(sets, out, sizes))[0]
        einstr = ",".join(terms) + "->" + out
        ops = [rng.standard_normal([sizes[c] for c in t]) for t in terms]
        clist = paths.contraction_list(einstr, [o.shape for o in ops], optimize=tuple(new))
        np.testing.assert_allclose(run_list(einstr, ops, clist), np.einsum(einstr, *ops), rtol=1e-9, atol=1e-9)
    assert improved > 15


def test_auto_on_a_lattice_beats_the_row_sweep():
    """8 x 8 PEPS, bond 8: 'auto' (4 noisy-greedy trials + subtree reconfiguration) ends below the hand-written
    row-by-row boundary sweep in multiply-adds, with the same largest intermediate, in about a second."""
    from contractn_amd import TN
    from contractn_amd.paths import ssa_to_linear
    from tests import networks as nets

    tn = nets.peps_closed(TN, 8, 8, 2, dtype=np.float32, seed=6)
    shapes = [tuple(8 if (d == 2 and p.ndim > 1 and ax > 0) else d for ax, d in enumerate(p.shape)) for p in tn.params]
    terms, out, sizes = paths.parse_einsum_input(tn.einsum_str, shapes)
    sets = [set(t) for t in terms]
    row = paths.path_cost(sets, out, sizes, ssa_to_linear(nets.peps_row_path(8, 8), 128))
    auto = paths.path_cost(sets, out, sizes, paths.find_path(terms, out, sizes, "auto"))
    greedy = paths.path_cost(sets, out, sizes, paths.find_path(terms, out, sizes, "greedy"))
    assert auto[0] < row[0] < greedy[0]
    assert auto[1] <= row[1]


def test_auto_finds_the_sweep_of_a_chain_on_a_batch_hyperedge():
    """BASELINE config 3b in small: an MPS whose every site also takes a batched input through one batch label.
    Pairwise greedy merges cores (or batch inputs); the single-cluster start walks the chain - `auto` must end at
    (or below) the hand-written left-to-right sweep."""
    from contractn_amd import TN
    from contractn_amd.paths import ssa_to_linear
    from tests import networks as nets

    class Shape:
        def __init__(self, shape):
            self.shape, self.ndim = tuple(shape), len(shape)

    batch, n_sites, bond, phys = 512, 20, 32, 4
    tn = TN()
    hub = tn.add_copy_node(n_sites + 1)
    cores = [Shape((phys, bond) if i in (0, n_sites - 1) else (phys, bond, bond)) for i in range(n_sites)]
    nodes = nets.add_mps(tn, cores)
    for i, node in enumerate(nodes):
        inp = tn.add_input_node((batch, phys), var_shape_axes=(0,))
        tn.connect_nodes(inp, node, 1, 0)
        tn.connect_nodes(hub, inp, i, 0)
    shapes = [c.shape for c in cores] + [(batch, phys)] * n_sites
    terms, out, sizes = paths.parse_einsum_input(tn.einsum_str, shapes)
    sets = [set(t) for t in terms]
    hand = paths.path_cost(sets, out, sizes, ssa_to_linear(nets.batched_mps_path(n_sites), 2 * n_sites))
    auto = paths.path_cost(sets, out, sizes, paths.find_path(terms, out, sizes, "auto"))
    greedy = paths.path_cost(sets, out, sizes, paths.find_path(terms, out, sizes, "greedy"))
    assert auto[0] <= hand[0] * 1.05 and auto[1] <= hand[1]
    assert greedy[0] > 2 * hand[0]          # what the start from pairwise greedy alone would have given
