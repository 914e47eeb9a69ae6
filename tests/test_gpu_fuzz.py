"""Randomised einsum structures on the GPU vs numpy.einsum (fp64, tight tolerance).

Exercises the label algebra of the planner: batch (kept+shared), free, contracted, summed-out and
diagonal labels, extent-1 axes, arbitrary output permutations, unary steps, multi-operand paths."""
import os

import numpy as np
import pytest

from contractn_amd import contract

# CTN_FUZZ_OFFSET=n shifts every seed: extended runs on a GPU box draw fresh cases without growing the suite
OFFSET = int(os.environ.get("CTN_FUZZ_OFFSET", "0"))

pytestmark = pytest.mark.gpu

LETTERS = "abcdefghij"


def random_pair_case(rng):
    n_lab = rng.integers(1, 7)
    labels = list(LETTERS[:n_lab])
    sizes = {l: int(rng.choice([1, 2, 3, 4, 5, 7, 8, 16, 33])) for l in labels}
    def term():
        k = rng.integers(0, min(n_lab, 4) + 1)
        t = list(rng.choice(labels, size=k, replace=False))
        if t and rng.random() < 0.15:           # diagonal: repeat one label
            t.insert(rng.integers(0, len(t) + 1), t[rng.integers(0, len(t))])
        return "".join(t)
    ta, tb = term(), term()
    present = sorted(set(ta + tb))
    keep = [l for l in present if rng.random() < 0.6]
    rng.shuffle(keep)
    return f"{ta},{tb}->{''.join(keep)}", sizes


@pytest.mark.parametrize("seed", range(120))
def test_random_pairwise_step(seed):
    rng = np.random.default_rng(OFFSET + seed)
    einstr, sizes = random_pair_case(rng)
    lhs = einstr.split("->")[0].split(",")
    ops = [rng.standard_normal([sizes[c] for c in t]) for t in lhs]
    ref = np.einsum(einstr, *ops)
    t_hat, c = contract(einstr, *ops, split_format=True)
    got = np.asarray(t_hat) * np.exp(float(c))
    assert got.shape == ref.shape, einstr
    scale = max(np.max(np.abs(ref)), 1e-300)
    assert np.max(np.abs(got - ref)) <= 1e-11 * scale, (einstr, sizes)


def random_network_case(rng):
    n_ops = int(rng.integers(3, 7))
    labels = list(LETTERS[: int(rng.integers(3, 8))])
    sizes = {l: int(rng.choice([2, 3, 4, 6])) for l in labels}
    terms = []
    for _ in range(n_ops):
        k = int(rng.integers(1, min(len(labels), 4) + 1))
        terms.append("".join(rng.choice(labels, size=k, replace=False)))
    present = sorted(set("".join(terms)))
    keep = [l for l in present if rng.random() < 0.35]
    rng.shuffle(keep)
    return ",".join(terms) + "->" + "".join(keep), sizes


def random_pair_f32_case(rng):
    pool = [1, 2, 4, 8, 12, 32, 40, 64, 100, 128]
    sizes = {l: int(rng.choice(pool)) for l in "abcde"}
    ta = "".join(rng.choice(list("abcde"), size=int(rng.integers(1, 4)), replace=False))
    tb = "".join(rng.choice(list("abcde"), size=int(rng.integers(1, 4)), replace=False))
    present = sorted(set(ta + tb))
    keep = [l for l in present if rng.random() < 0.6]
    rng.shuffle(keep)
    return f"{ta},{tb}->{''.join(keep)}", sizes


@pytest.mark.parametrize("seed", range(40))
def test_random_network(seed):
    """3-6 operands, random shared labels (hyperedges allowed), auto path."""
    rng = np.random.default_rng(OFFSET + 1000 + seed)
    einstr, sizes = random_network_case(rng)
    terms = einstr.split("->")[0].split(",")
    ops = [rng.standard_normal([sizes[c] for c in t]) for t in terms]
    ref = np.einsum(einstr, *ops)
    t_hat, c = contract(einstr, *ops, split_format=True)
    got = np.asarray(t_hat) * np.exp(float(c))
    scale = max(np.max(np.abs(ref)), 1e-300)
    assert np.max(np.abs(got - ref)) <= 1e-10 * scale, einstr


@pytest.mark.parametrize("seed", range(30))
def test_random_pairwise_step_f32_larger(seed):
    """fp32 with extents that reach the MFMA / row-dot / vector-stream kernels."""
    rng = np.random.default_rng(OFFSET + 5000 + seed)
    einstr, sizes = random_pair_f32_case(rng)
    ta, tb = einstr.split("->")[0].split(",")
    ops = [rng.standard_normal([sizes[c] for c in t]).astype(np.float32) for t in (ta, tb)]
    ref = np.einsum(einstr, *[o.astype(np.float64) for o in ops])
    t_hat, c = contract(einstr, *ops, split_format=True)
    got = np.asarray(t_hat, dtype=np.float64) * np.exp(float(c))
    scale = max(np.max(np.abs(ref)), 1e-300)
    # a sum that cancels (e.g. 'ec,a->' over zero-mean data) is only accurate relative to its TERMS in fp32
    terms = np.max(np.einsum(einstr, *[np.abs(o).astype(np.float64) for o in ops]))
    assert np.max(np.abs(got - ref)) <= 2e-4 * scale + 2e-6 * terms, (einstr, sizes)


def _large_tile_case(rng, dtype="float32"):
    """Random two-operand step in the shape class of the large-tile LDS-DMA kernels: every operand unit-stride
    along its free labels or along k.  Returns (einsum, sizes)."""
    from contractn_amd import einsum as E
    for _ in range(5000):
        nm, nn, nk = int(rng.integers(1, 3)), int(rng.integers(1, 3)), int(rng.integers(1, 3))
        m_l, n_l, k_l = list("ab"[:nm]), list("cd"[:nn]), list("ef"[:nk])
        batch = ["x"] if rng.random() < 0.3 else []
        sizes = {l: int(rng.choice([4, 8, 12, 16, 20, 32, 64])) for l in m_l + n_l}
        sizes.update({l: int(rng.choice([2, 3, 4, 5, 8, 11, 16, 32])) for l in k_l})
        sizes.update({l: int(rng.choice([2, 3])) for l in batch})
        ka, kb = list(k_l), list(k_l)
        rng.shuffle(ka); rng.shuffle(kb)
        # each operand either unit-stride along its free labels (k first) or along k (k last)
        ta = "".join(batch + (ka + m_l if rng.random() < 0.5 else m_l + ka))
        tb = "".join(batch + (kb + n_l if rng.random() < 0.5 else n_l + kb))
        out_m = list(m_l); rng.shuffle(out_m)                   # row labels of C in any order
        einstr = f"{ta},{tb}->{''.join(batch + out_m + n_l)}"
        shapes = tuple(tuple(sizes[c] for c in t) for t in (ta, tb))
        clist = E._contract_path(einstr, shapes, optimize=((0, 1),), memory_limit=None, use_blas=True)
        info = E._native_plan(clist, shapes, dtype).step_infos()[0]
        if (info["kernel"] == 2 and info["tile_m"] == 256) or (info["kernel"] == 3 and info["tile_n"] == 128):
            return einstr, sizes
    raise AssertionError("no eligible case generated")


@pytest.mark.parametrize("seed", range(40))
def test_random_large_tile_steps_f32(seed, monkeypatch):
    from contractn_amd import einsum as E
    monkeypatch.setenv("CTN_MFMA_G", "2")       # take the large-tile kernel whenever the planner allows it
    E.clear_caches()
    rng = np.random.default_rng(OFFSET + 9000 + seed)
    einstr, sizes = _large_tile_case(rng)
    lhs = einstr.split("->")[0].split(",")
    ops = [rng.standard_normal([sizes[c] for c in t]).astype(np.float32) for t in lhs]
    ref = np.einsum(einstr, *[o.astype(np.float64) for o in ops])
    t_hat, c = contract(einstr, *ops, optimize=((0, 1),), split_format=True)
    E.clear_caches()
    got = t_hat.astype(np.float64) * np.exp(float(c))
    assert got.shape == ref.shape, einstr
    assert np.max(np.abs(got - ref)) <= 1e-4 * np.max(np.abs(ref)), (einstr, sizes)


@pytest.mark.parametrize("seed", range(16))
def test_random_large_tile_steps_f64(seed, monkeypatch):
    from contractn_amd import einsum as E
    monkeypatch.setenv("CTN_MFMA_G", "2")
    E.clear_caches()
    rng = np.random.default_rng(OFFSET + 9500 + seed)
    einstr, sizes = _large_tile_case(rng, "float64")
    lhs = einstr.split("->")[0].split(",")
    ops = [rng.standard_normal([sizes[c] for c in t]) for t in lhs]
    ref = np.einsum(einstr, *ops)
    t_hat, c = contract(einstr, *ops, optimize=((0, 1),), split_format=True)
    E.clear_caches()
    got = t_hat * np.exp(float(c))
    assert got.shape == ref.shape, einstr
    assert np.max(np.abs(got - ref)) <= 1e-11 * np.max(np.abs(ref)), (einstr, sizes)


def _epilogue_sum_case(rng):
    """Random instance of planner pattern C: a GEMM (row labels from x, column labels from the core, something summed)
    whose result a network INPUT re-weights and sums over one short column label.  Axis orders of x, of the core and
    of the output are random; the weights carry a random non-empty subset of the row labels and end in the short
    label.  Returns (einsum, sizes, short label)."""
    from contractn_amd import einsum as E
    for _ in range(2000):
        rows = list("ab"[:int(rng.integers(1, 3))])
        cols = list("cd"[:int(rng.integers(1, 3))])
        ks = list("ef"[:int(rng.integers(1, 3))])
        batch = ["x"] if rng.random() < 0.25 else []
        sizes = {l: int(rng.choice([8, 12, 16, 24, 32, 50, 64, 96])) for l in rows + cols}
        sizes.update({l: int(rng.choice([4, 8, 12, 16, 33])) for l in ks})
        sizes.update({l: int(rng.choice([2, 3])) for l in batch})
        sizes["p"] = int(rng.choice([2, 4]))
        m = int(np.prod([sizes[l] for l in rows])); n = int(np.prod([sizes[l] for l in cols])) * sizes["p"]
        k = int(np.prod([sizes[l] for l in ks])); b = int(np.prod([sizes[l] for l in batch])) if batch else 1
        if m < 64 or n < 32 or k < 8 or b * m * n < 65536 or b * m * n > (1 << 22):
            continue
        tx = batch + rows + ks; rng.shuffle(tx)
        tc = batch + cols + ks + ["p"]; rng.shuffle(tc)
        wrows = [l for l in rows if rng.random() < 0.7] or [rows[0]]
        rng.shuffle(wrows)
        tw = wrows + ["p"]
        out = batch + rows + cols; rng.shuffle(out)
        einstr = f"{''.join(tx)},{''.join(tc)},{''.join(tw)}->{''.join(out)}"
        shapes = tuple(tuple(sizes[c] for c in t) for t in (tx, tc, tw))
        clist = E._contract_path(einstr, shapes, optimize=((0, 1), (0, 1)), memory_limit=None, use_blas=True)
        infos = E._native_plan(clist, shapes, "float32").step_infos()
        if infos[0]["kernel"] == 5 and infos[1]["epilogue_sum"] == sizes["p"]:
            return einstr, sizes
    raise AssertionError("no epilogue-sum case generated")


@pytest.mark.parametrize("seed", range(30))
def test_random_epilogue_sum_steps_f32(seed):
    from contractn_amd import einsum as E
    E.clear_caches()
    rng = np.random.default_rng(OFFSET + 13000 + seed)
    einstr, sizes = _epilogue_sum_case(rng)
    lhs = einstr.split("->")[0].split(",")
    ops = [(rng.standard_normal([sizes[c] for c in t]) * rng.uniform(0.3, 3.0)).astype(np.float32) for t in lhs]
    ref = np.einsum(einstr, *[o.astype(np.float64) for o in ops])
    t_hat, c = contract(einstr, *ops, optimize=((0, 1), (0, 1)), split_format=True)
    got = np.asarray(t_hat, dtype=np.float64) * np.exp(float(c))
    terms = np.max(np.einsum(einstr, *[np.abs(o).astype(np.float64) for o in ops]))
    assert got.shape == ref.shape
    assert np.max(np.abs(got - ref)) <= 2e-4 * np.max(np.abs(ref)) + 2e-6 * terms, (einstr, sizes)
    assert abs(np.mean(np.abs(t_hat)) - 1.0) < 1e-5
