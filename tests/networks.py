"""Builders for the benchmark / parity networks (shared by tests, bench.py and
oracle/gen_golden.py).  Every builder takes the TN class to use, so the same
code drives the reference's ``contractn.TN`` (golden generation, build
container only) and ``contractn_amd.TN``.  Networks are built through the
public TN API exactly like reference contractn/tests/test_einsum.py:41-52.
"""
import numpy as np


def _rand(rng, shape, scale, dtype):
    # cast AFTER scaling (NEP-50: float32 / np.float64 scalar -> float64)
    return (rng.standard_normal(shape) / scale).astype(dtype)


def mps_cores(n_sites, bond, phys, dtype, seed, scale=None):
    """Open-boundary MPS cores, index order (phys, left, right)."""
    rng = np.random.default_rng(seed)
    scale = np.sqrt(bond) if scale is None else scale
    cores = []
    for i in range(n_sites):
        if i == 0 or i == n_sites - 1:
            shape = (phys, bond)
        else:
            shape = (phys, bond, bond)
        cores.append(_rand(rng, shape, scale, dtype))
    return cores


def add_mps(tn, cores):
    nodes = [tn.add_dense_node(c) for c in cores]
    n = len(nodes)
    for i in range(n - 1):
        tn.connect_nodes(nodes[i], nodes[i + 1], -1, -2 if i != n - 2 else -1)
    return nodes


def mps_overlap(TN, n_sites, bond, phys, dtype=np.float32, seed=3, scale=None):
    """<phi|psi> of two open-boundary MPS (BASELINE config 3a).  Returns (tn, ssa zipper path)."""
    psi = mps_cores(n_sites, bond, phys, dtype, seed, scale)
    phi = mps_cores(n_sites, bond, phys, dtype, seed + 1000, scale)
    tn = TN()
    psi_nodes = add_mps(tn, psi)
    phi_nodes = add_mps(tn, phi)
    for a, b in zip(psi_nodes, phi_nodes):
        tn.connect_nodes(a, b, 0, 0)
    return tn, zipper_path(n_sites)


def zipper_path(n_sites):
    """SSA path: E=(psi0,phi0); then for every site (E,psi_i) -> T, (T,phi_i) -> E."""
    n = n_sites
    path = [(0, n)]
    cur = 2 * n
    for i in range(1, n):
        path.append((cur, i))
        path.append((cur + 1, n + i))
        cur += 2
    return path


def mps_open(TN, bonds, phys, dtype=np.float64, seed=11, ones=False):
    """Open MPS with free physical legs (reference tests/test_einsum.py:28-64 topology)."""
    rng = np.random.default_rng(seed)
    n = len(phys)
    tn = TN()
    cores = []
    for i in range(n):
        if i == 0:
            shape = (phys[0], bonds[0])
        elif i == n - 1:
            shape = (phys[-1], bonds[-1])
        else:
            shape = (phys[i], bonds[i - 1], bonds[i])
        cores.append(np.ones(shape, dtype=dtype) if ones else _rand(rng, shape, 1.0, dtype))
    add_mps(tn, cores)
    return tn


def cp_network(TN, rank, dims, dtype=np.float64, seed=5, scale=1.0):
    """CP decomposition: copy-node hub + factor matrices (rank, n_i).  'ac,ad,ae->cde'."""
    rng = np.random.default_rng(seed)
    tn = TN()
    hub = tn.add_copy_node(len(dims))
    for i, n in enumerate(dims):
        mat = tn.add_dense_node(_rand(rng, (rank, n), scale, dtype))
        tn.connect_nodes(hub, mat, i, 0)
    return tn


def tucker_network(TN, ranks, dims, dtype=np.float64, seed=5, scale=1.0, delta_hub=False):
    """Tucker: dense hub + factor matrices.  ``delta_hub`` materialises the copy tensor
    (must equal the CP network built from the same seed)."""
    rng = np.random.default_rng(seed)
    tn = TN()
    mats = [_rand(rng, (r, n), scale, dtype) for r, n in zip(ranks, dims)]
    if delta_hub:
        assert len(set(ranks)) == 1
        hub_t = np.zeros(ranks, dtype=dtype)
        for i in range(ranks[0]):
            hub_t[(i,) * len(ranks)] = 1
    else:
        hub_t = _rand(rng, tuple(ranks), scale, dtype)
    hub = tn.add_dense_node(hub_t)
    for i, m in enumerate(mats):
        node = tn.add_dense_node(m)
        tn.connect_nodes(hub, node, i, 0)
    return tn


def peps_closed(TN, rows, cols, bond, phys=2, dtype=np.float32, seed=6):
    """rows x cols PEPS; site legs (phys, up, left, down, right) - only the existing ones -
    closed with one vector per physical leg (BASELINE config 5 rendition)."""
    rng = np.random.default_rng(seed)
    tn = TN()
    site = {}
    legs = {}
    for r in range(rows):
        for c in range(cols):
            names = ["p"]
            if r > 0:
                names.append("u")
            if c > 0:
                names.append("l")
            if r < rows - 1:
                names.append("d")
            if c < cols - 1:
                names.append("r")
            shape = tuple(phys if n == "p" else bond for n in names)
            site[r, c] = tn.add_dense_node(_rand(rng, shape, np.sqrt(bond), dtype))
            legs[r, c] = names
    for r in range(rows):
        for c in range(cols):
            if c < cols - 1:
                tn.connect_nodes(site[r, c], site[r, c + 1], legs[r, c].index("r"), legs[r, c + 1].index("l"))
            if r < rows - 1:
                tn.connect_nodes(site[r, c], site[r + 1, c], legs[r, c].index("d"), legs[r + 1, c].index("u"))
    for r in range(rows):
        for c in range(cols):
            vec = tn.add_dense_node(_rand(rng, (phys,), 1.0, dtype))
            tn.connect_nodes(site[r, c], vec, 0, 0)
    return tn


def batched_mps(TN, n_sites, bond, phys, batch, dtype=np.float32, seed=4):
    """One MPS evaluated on a batch of product inputs through a batch hyperedge
    (BASELINE config 3b).  The copy node is created first.
    Returns (tn, inputs)."""
    rng = np.random.default_rng(seed)
    tn = TN()
    hub = tn.add_copy_node(n_sites + 1)
    cores = mps_cores(n_sites, bond, phys, dtype, seed)
    nodes = add_mps(tn, cores)
    inputs = []
    for i, node in enumerate(nodes):
        inp = tn.add_input_node((batch, phys), var_shape_axes=(0,))
        tn.connect_nodes(inp, node, 1, 0)
        tn.connect_nodes(hub, inp, i, 0)
        inputs.append(_rand(rng, (batch, phys), 1.0, dtype))
    return tn, inputs


def peps_row_path(rows, cols):
    """SSA path for peps_closed(): absorb every physical vector into its site, then sweep the
    sites row by row into one boundary tensor (largest intermediate: bond ** (cols + 1))."""
    n = rows * cols
    path = [(i, n + i) for i in range(n)]          # site_i x vec_i -> id 2n + i
    cur = 2 * n
    nxt = 3 * n
    for i in range(1, n):
        path.append((cur, 2 * n + i))
        cur = nxt
        nxt += 1
    return path


def batched_mps_path(n_sites):
    """SSA sweep for batched_mps(): per site one GEMM (batch x bond) . (bond x phys*bond) followed by
    the hyperedge step 'apr,ap->ar' that feeds the batched input in."""
    n = n_sites
    path = [(0, n)]
    cur, nxt = 2 * n, 2 * n + 1
    for i in range(1, n):
        path.append((cur, i))
        path.append((nxt, n + i))
        cur, nxt = nxt + 1, nxt + 2
    return path
