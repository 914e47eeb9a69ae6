"""The arithmetic behind k_sweep_f32's stabilisation (contractn_amd/csrc/kernels_sweep.h), restated in NumPy and checked
against the oracle on the CPU: blocks of rows that rescale by their OWN mean and only record (abs-sum, applied scale)
per site carry enough to reconstruct the rescale factor the reference applies to the WHOLE tensor after every step
(reference einsum.py:97-106), and the final stored tensor."""
import numpy as np
import pytest

from oracle import cpu_ref


def _chain(rng, batch, bond, phys, sites, spread):
    cores = [(rng.standard_normal((bond, phys, bond)) / np.sqrt(bond)).astype(np.float64) for _ in range(sites)]
    xs = [rng.standard_normal((batch, phys)) * 0.7 for _ in range(sites)]
    for x in xs:
        x[batch // 2:] *= spread ** (1.0 / sites)         # half of the inputs grow much faster than the rest
    e0 = rng.standard_normal((batch, bond))
    return e0, cores, xs


def _reference(e0, cores, xs):
    """The reference's loop on the whole tensor: two pairwise steps per site, stabilize after each."""
    t, log_scale, rescales = e0, np.zeros(()), []
    t, log_scale = cpu_ref.stabilize(t, log_scale)
    for w, x in zip(cores, xs):
        c = np.einsum("bl,lpr->bpr", t, w)
        # (the engine fuses the two steps: the intermediate is not rescaled, its magnitude moves into the second step)
        nxt = np.einsum("bpr,bp->br", c, x)
        norm = np.sum(np.abs(nxt))
        rescales.append(norm / nxt.size if norm > cpu_ref.MIN_NORM else 0.0)
        t, log_scale = cpu_ref.stabilize(nxt, log_scale)
    return t, float(log_scale), np.array(rescales)


def _sweep(e0, cores, xs, rows):
    """What the kernels do: kernels_sweep.h k_sweep_f32 (per block), k_sweep_logs / k_sweep_z / k_sweep_finish."""
    batch, bond = e0.shape
    J, S = batch // rows, len(cores)
    rec_a, rec_s = np.zeros((S, J)), np.ones((S, J))
    e_in, _ = cpu_ref.stabilize(e0, np.zeros(()))
    out = np.zeros_like(e0)
    for j in range(J):
        state, inv = e_in[j * rows:(j + 1) * rows].copy(), 1.0
        for s in range(S):
            x = xs[s][j * rows:(j + 1) * rows] * inv                      # the scale goes into the weights
            state = np.einsum("bpr,bp->br", np.einsum("bl,lpr->bpr", state, cores[s]), x)
            a = np.sum(np.abs(state))
            sc = a / state.size if (s + 1 < S and a > 1e-30) else 1.0
            rec_a[s, j], rec_s[s, j] = a, sc
            inv = 1.0 / sc
        out[j * rows:(j + 1) * rows] = state                              # leaves with the block's own scale
    la = np.where(rec_a > 0, np.log(np.where(rec_a > 0, rec_a, 1.0)), -np.inf)
    ls = np.log(rec_s)
    g_before = np.vstack([np.zeros((1, J)), np.cumsum(ls, axis=0)[:-1]])      # sum_{i < s} log s[i][j]
    t = la + g_before
    mx = t.max(axis=1)
    Z = mx + np.log(np.sum(np.exp(t - mx[:, None]), axis=1)) - np.log(e0.size)
    log_r, norms = 0.0, []
    for s in range(S):
        ln = np.log(e0.size) + Z[s] - log_r
        log_r_before = log_r
        norms.append(np.exp(ln))
        if ln > np.log(cpu_ref.MIN_NORM):
            log_r = Z[s]
    fac = np.exp(g_before[S - 1] - log_r_before)                           # per block: exp(g[j][S-2]) / R_{S-2}
    stored = out * np.repeat(fac, rows)[:, None]
    return stored, np.array(norms) / e0.size


@pytest.mark.parametrize("spread", [1.0, 1e6, 1e-5])
def test_block_local_rescale_reconstructs_the_reference_rescale_factors(spread):
    rng = np.random.default_rng(3)
    e0, cores, xs = _chain(rng, batch=96, bond=24, phys=3, sites=9, spread=spread)
    t_ref, c_ref, resc_ref = _reference(e0, cores, xs)
    stored, resc = _sweep(e0, cores, xs, rows=16)
    assert np.allclose(resc, resc_ref, rtol=1e-12, atol=0.0)
    # the stored tensor of the last step is the reference's pre-stabilize tensor: dividing by its mean gives T_hat
    assert np.allclose(stored / (np.sum(np.abs(stored)) / stored.size), t_ref, rtol=1e-10, atol=1e-300)
    assert abs(np.sum(np.log(resc)) + np.log(np.sum(np.abs(e0)) / e0.size) - c_ref) <= 1e-10 * max(1.0, abs(c_ref))
