"""Generate tests/golden/*.npz by running the UNMODIFIED reference in the build container.

Run (build container only - /root/reference does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=oracle/ref_shim:/root/reference \
        python oracle/gen_golden.py

Everything numerical is produced by the reference's own files
(contractn/ctn.py, nodes.py, einsum.py: TN API, make_einstring, contract,
_core_contract, stabilize) on NumPy; ``oracle/ref_shim/opt_einsum`` only supplies
the third-party names the reference imports (path bookkeeping + NumPy
dispatch), and every fixture pins an EXPLICIT path so both sides run the same DAG.

A fixture is data only: einsum string, path, input arrays, expected outputs.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import contractn  # the reference (via PYTHONPATH)  # noqa: E402
from contractn import TN  # noqa: E402

assert contractn.__file__.startswith("/root/reference"), contractn.__file__

from contractn_amd.paths import contraction_list, find_path, parse_einsum_input, ssa_to_linear  # noqa: E402
from tests import networks as nets  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def own_path(einstr, shapes, optimize="greedy"):
    terms, out, sizes = parse_einsum_input(einstr, shapes)
    return tuple(tuple(p) for p in find_path(terms, out, sizes, optimize))


ONLY = set(sys.argv[1:])   # fixture names to (re)generate; none = all


def save(name, tn, path, inputs=(), note=""):
    if ONLY and name not in ONLY:
        return
    einstr = tn.einsum_str
    params = tn.params
    np.seterr(all="ignore")
    fun_split = tn.make_contract_fun(optimize=path, split_format=True)
    fun_plain = tn.make_contract_fun(optimize=path, split_format=False)
    t_hat, log_scale = fun_split(params, inputs)
    plain = fun_plain(params, inputs)
    # operand tuple exactly as the reference packs it
    from contractn.einsum import make_arg_packer

    operands = make_arg_packer(tn)(params, inputs)
    data = {
        "einsum_str": np.array(einstr),
        "path": np.array([list(p) + [-1] * (2 - len(p)) for p in path], dtype=np.int64),
        "n_operands": np.array(len(operands)),
        "t_hat": np.asarray(t_hat),
        "log_scale": np.asarray(log_scale, dtype=np.float64),
        "log_scale_hex": np.array(float(log_scale).hex()),
        "plain": np.asarray(plain),
        "note": np.array(note),
    }
    for i, op in enumerate(operands):
        data[f"op{i}"] = np.asarray(op)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **data)
    print(f"{name:28s} {einstr[:40]:40s} out{np.shape(t_hat)} dtype={np.asarray(t_hat).dtype} "
          f"log_scale={float(log_scale):+.12g}")


def save_torch(name, tn, path, note=""):
    """The same network through the reference's TORCH backend (einsum.py:9-21: CPU tensors here): the register is a
    ``torch.float32`` 0-d tensor accumulated in fp32 (``torch.zeros(())``, einsum.py:338; SURVEY.md App. A last row),
    which the NumPy fixtures cannot pin.  Operands: the TN's tensors as float32."""
    if ONLY and name not in ONLY:
        return
    import torch
    from contractn import contract as ref_contract
    from contractn.einsum import make_arg_packer

    operands = [np.asarray(o, dtype=np.float32) for o in make_arg_packer(tn)(tn.params, ())]
    t_ops = [torch.from_numpy(o) for o in operands]
    t_hat, log_scale = ref_contract(tn.einsum_str, *t_ops, optimize=path, split_format=True)
    plain = ref_contract(tn.einsum_str, *t_ops, optimize=path, split_format=False)
    assert log_scale.dtype == torch.float32 and t_hat.dtype == torch.float32
    data = {
        "einsum_str": np.array(tn.einsum_str),
        "path": np.array([list(p) + [-1] * (2 - len(p)) for p in path], dtype=np.int64),
        "n_operands": np.array(len(operands)),
        "t_hat": t_hat.numpy(),
        "log_scale": np.asarray(log_scale.numpy(), dtype=np.float32),
        "log_scale_hex": np.array(float(log_scale).hex()),
        "plain": plain.numpy(),
        "note": np.array(note),
    }
    for i, op in enumerate(operands):
        data[f"op{i}"] = op
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **data)
    print(f"{name:28s} torch fp32 register log_scale={float(log_scale):+.9g} ({float(log_scale).hex()})")


def main():
    os.makedirs(OUT, exist_ok=True)

    # -- README example 1: copy tensor of order 101 (README.md:18-34)
    tn = TN()
    hub = tn.add_copy_node(101)
    for i in range(100):
        vec = tn.add_dense_node(np.array([1, 0.99]))
        tn.connect_nodes(hub, vec, i, 0)
    path = tuple(nets_left_to_right(100))
    save("readme_copy101", tn, path, note="README.md:18-34; expected [1, 0.99**100]")

    # -- README example 3: 1000-matrix chain (README.md:62-77)
    tn = TN()
    prev = tn.add_dense_node(np.ones((3,)))
    for _ in range(1000):
        mat = tn.add_dense_node(np.ones((3, 3)))
        tn.connect_nodes(prev, mat, -1, 0)
        prev = mat
    save("readme_chain1000", tn, tuple(nets_left_to_right(1001)),
         note="README.md:62-77; split -> ([1,1,1], 1098.61228867), plain -> inf")
    save_torch("readme_chain1000_torch_f32", tn, tuple(nets_left_to_right(1001)),
               note="README.md:62-77 on the torch backend: 1000 x log(3) added up in fp32")

    # -- random chain (config 2 random variant)
    rng = np.random.default_rng(2)
    tn = TN()
    prev = tn.add_dense_node(rng.uniform(0.5, 1.5, (3,)))
    for _ in range(200):
        mat = tn.add_dense_node(rng.uniform(0.5, 1.5, (3, 3)))
        tn.connect_nodes(prev, mat, -1, 0)
        prev = mat
    save("chain200_random", tn, tuple(nets_left_to_right(201)), note="U(0.5,1.5) entries, seed 2")

    # -- MPS overlap, zipper path (config 3a, reduced size)
    for dtype, tag in ((np.float64, "f64"), (np.float32, "f32")):
        tn, ssa = nets.mps_overlap(TN, 6, 8, 3, dtype=dtype, seed=3)
        save(f"mps_overlap_6x8x3_{tag}", tn, ssa_to_linear(ssa, 12), note="zipper path")
    tn, ssa = nets.mps_overlap(TN, 5, 64, 4, dtype=np.float32, seed=3)
    save("mps_overlap_5x64x4_f32", tn, ssa_to_linear(ssa, 10), note="zipper path; MFMA-sized steps")
    save_torch("mps_overlap_5x64x4_torch_f32", tn, ssa_to_linear(ssa, 10), note="zipper path, torch backend")
    tn, ssa = nets.mps_overlap(TN, 4, 48, 4, dtype=np.float64, seed=4)
    save("mps_overlap_4x48x4_f64", tn, ssa_to_linear(ssa, 8), note="zipper path")

    # -- the same zipper with operands of extreme magnitude: every NORMALISED quantity stays in range (the reference
    #    rescales each intermediate before the next product, einsum.py:387), products of un-normalised ones do not
    #    (fp32: 4 x^2 * x * sqrt(K) leaves the range for x = 1e13 and sinks into the subnormals for x = 1e-14)
    for mag, tag in ((1e13, "huge"), (1e-14, "tiny")):
        tn, ssa = nets.mps_overlap(TN, 4, 32, 4, dtype=np.float32, seed=3, scale=1.0 / mag)
        save(f"mps_overlap_4x32x4_f32_{tag}", tn, ssa_to_linear(ssa, 8), note=f"zipper path; entries ~ {mag:g}")

    # -- complex tensors (the reference contracts them through NumPy: modulus norm, real register)
    def complexify(tn, seed, cdtype, every=1):
        rng_c = np.random.default_rng(seed)
        for k, node in enumerate(tn.nodes(as_iter=True, copy_nodes=False, danglers=False)):
            if k % every == 0:
                t = np.asarray(node.tensor)
                node.tensor = (t + 1j * rng_c.standard_normal(t.shape) * np.mean(np.abs(t))).astype(cdtype)

    tn, ssa = nets.mps_overlap(TN, 5, 12, 3, dtype=np.float64, seed=8)
    complexify(tn, 80, np.complex128)
    save("mps_overlap_5x12x3_c128", tn, ssa_to_linear(ssa, 10), note="zipper path; complex128 cores")
    tn, ssa = nets.mps_overlap(TN, 4, 40, 4, dtype=np.float32, seed=9)
    complexify(tn, 81, np.complex64)
    save("mps_overlap_4x40x4_c64", tn, ssa_to_linear(ssa, 8), note="zipper path; complex64 cores, MFMA-sized steps")
    tn, ssa = nets.mps_overlap(TN, 4, 10, 3, dtype=np.float64, seed=10)
    complexify(tn, 82, np.complex128, every=2)
    save("mps_overlap_4x10x3_mixed_c128", tn, ssa_to_linear(ssa, 8), note="zipper path; every second core complex, the rest real")
    tn = nets.mps_open(TN, (3, 5, 4), (2, 3, 2, 4), dtype=np.float64, seed=12)
    complexify(tn, 83, np.complex128)
    save("mps_open_random_c128", tn, own_path(tn.einsum_str, [p.shape for p in tn.params]), note="open legs, complex128")
    tn = nets.cp_network(TN, 5, (6, 7, 8), dtype=np.float64, seed=13)
    complexify(tn, 84, np.complex128)
    save("cp_r5_c128", tn, own_path(tn.einsum_str, [p.shape for p in tn.params]), note="ac,ad,ae->cde, complex128: hyperedge")

    # -- open MPS, random and all-ones (reference tests/test_einsum.py:28-64)
    tn = nets.mps_open(TN, (3, 5, 4), (2, 3, 2, 4), dtype=np.float64, seed=11)
    ein = tn.einsum_str
    save("mps_open_random_f64", tn, own_path(ein, [p.shape for p in tn.params]), note="open legs")
    tn = nets.mps_open(TN, (7, 2, 10, 3), (3, 1, 6, 2, 5), dtype=np.float64, ones=True)
    ein = tn.einsum_str
    save("mps_open_ones_f64", tn, own_path(ein, [p.shape for p in tn.params]),
         note="all ones: every entry == prod(bond dims)")

    # -- CP (hyperedge) vs Tucker (dense hub / materialised delta hub) (config 4, reduced)
    for dtype, tag in ((np.float64, "f64"), (np.float32, "f32")):
        tn = nets.cp_network(TN, 5, (6, 7, 8), dtype=dtype, seed=5)
        save(f"cp_r5_{tag}", tn, own_path(tn.einsum_str, [p.shape for p in tn.params]), note="ac,ad,ae->cde")
    tn = nets.tucker_network(TN, (5, 5, 5), (6, 7, 8), dtype=np.float64, seed=5)
    save("tucker_r5_f64", tn, own_path(tn.einsum_str, [p.shape for p in tn.params]), note="abc,ae,bf,cg->efg")
    tn = nets.tucker_network(TN, (5, 5, 5), (6, 7, 8), dtype=np.float64, seed=5, delta_hub=True)
    save("tucker_delta_r5_f64", tn, own_path(tn.einsum_str, [p.shape for p in tn.params]),
         note="delta hub: must equal cp_r5_f64")
    tn = nets.cp_network(TN, 48, (40, 36, 44), dtype=np.float32, seed=5, scale=4.0)
    save("cp_r48_f32", tn, own_path(tn.einsum_str, [p.shape for p in tn.params]), note="MFMA-sized")

    # -- PEPS 3x3 closed (config 5, reduced)
    for dtype, tag in ((np.float64, "f64"), (np.float32, "f32")):
        tn = nets.peps_closed(TN, 3, 3, 2, dtype=dtype, seed=7)
        save(f"peps3x3_D2_{tag}", tn, own_path(tn.einsum_str, [p.shape for p in tn.params]), note="greedy path")
    tn = nets.peps_closed(TN, 3, 4, 3, dtype=np.float64, seed=6)
    save("peps3x4_D3_f64", tn, own_path(tn.einsum_str, [p.shape for p in tn.params]), note="greedy path")

    # -- batched inputs through a batch hyperedge (config 3b, reduced)
    tn, inputs = nets.batched_mps(TN, 5, 6, 3, 7, dtype=np.float64, seed=4)
    shapes = None
    from contractn.einsum import make_arg_packer

    ops = make_arg_packer(tn)(tn.params, inputs)
    save("batched_mps_f64", tn, own_path(tn.einsum_str, [o.shape for o in ops]), inputs=inputs,
         note="input nodes + batch copy node")

    # -- edge cases
    tn = TN()
    a = tn.add_dense_node(np.eye(3))
    tn.connect_nodes(a, a, 0, 1)
    save("edge_trace", tn, ((0,),), note="aa-> ; 3.0000000000000004")
    tn = TN()
    tn.add_dense_node(np.arange(1.0, 7.0).reshape(2, 3))
    save("edge_single_node", tn, ((0,),), note="ab->ab single operand")
    tn = TN()
    tn.add_dense_node(np.array([1.0, -2.0]))
    tn.add_dense_node(np.array([3.0, 0.5, -1.0]))
    save("edge_outer", tn, ((0, 1),), note="a,b->ab outer product")
    tn = TN()
    x = tn.add_dense_node(np.zeros((2, 3)))
    y = tn.add_dense_node(np.ones((3, 2)))
    tn.connect_nodes(x, y, 1, 0)
    save("edge_zero", tn, ((0, 1),), note="zero tensor: unchanged, log_scale 0")
    rng = np.random.default_rng(9)
    tn = TN()
    x = tn.add_dense_node(rng.standard_normal((4, 3, 5)))
    y = tn.add_dense_node(rng.standard_normal((5, 2)))
    tn.connect_nodes(x, y, 2, 0)
    save("edge_sumout_transpose", tn, ((0, 1),), note="abc,cd->abd with odd sizes")


def nets_left_to_right(n):
    path = [(0, 1)] + [(0, n - 2 - k) for k in range(n - 2)]
    return path if n > 1 else [(0,)]


if __name__ == "__main__":
    main()
