"""The host planner (contractn_amd/csrc/plan.cpp) under AddressSanitizer + UBSan, on the CPU.

`make -C contractn_amd/csrc asan` builds plan.cpp with a small host-only driver (plan_check.cpp) under
``-fsanitize=address,undefined``.  This test feeds it every plan the suites exercise - the golden fixtures with
their explicit paths, the fuzz generators' structures (diagonals, summed-out labels, extent-1 axes, hyperedges,
the large-tile shape class), the headline 100-site network, PEPS sweeps and the 1024-wide CP / Tucker networks -
and requires: no sanitizer report, every gather-table entry inside the tensor it indexes, and the same flop count
as the production build of the planner (the -O3 object inside libctn_hip.so)."""
import os
import subprocess

import numpy as np
import pytest

from contractn_amd import TN
from contractn_amd import einsum as E
from contractn_amd.paths import ssa_to_linear
from tests import networks as nets
from tests.helpers import ROOT, golden_names, load_golden

CSRC = os.path.join(ROOT, "contractn_amd", "csrc")
BINARY = os.path.join(ROOT, "contractn_amd", "lib", "plan_check_asan")


def describe(einstr, shapes, optimize, dtype):
    """(text for plan_check, flops of the production planner) for one network."""
    shapes = tuple(tuple(int(d) for d in s) for s in shapes)
    clist = E._contract_path(einstr, shapes, optimize=optimize, memory_limit=None, use_blas=True)
    in_labels, steps = E.lower_contraction_list(len(shapes), clist)
    lines = [f"plan {1 if np.dtype(dtype) == np.float64 else 0} {len(shapes)} {len(steps)}"]
    for lab, shp in zip(in_labels, shapes):
        lines.append(" ".join(["in", str(len(shp))] + [str(d) for d in shp] + [str(x) for x in lab]))
    for lhs, rhs, out in steps:
        lines.append(" ".join(["step", str(lhs), str(rhs), str(len(out))] + [str(x) for x in out]))
    plan = E._native_plan(clist, shapes, np.dtype(dtype).name)
    return "\n".join(lines), plan.flops


def all_cases():
    from tests import test_gpu_fuzz as fz

    cases = []
    for name in golden_names():
        g = load_golden(name)
        dt = np.result_type(*[o.dtype for o in g["operands"]])
        dt = np.float32 if dt == np.float32 else np.float64
        cases.append((name, g["einsum_str"], [o.shape for o in g["operands"]], g["path"], dt))
    for seed in range(120):
        einstr, sizes = fz.random_pair_case(np.random.default_rng(seed))
        cases.append((f"pair{seed}", einstr, [[sizes[c] for c in t] for t in einstr.split("->")[0].split(",")], "auto", np.float64))
    for seed in range(40):
        einstr, sizes = fz.random_network_case(np.random.default_rng(1000 + seed))
        cases.append((f"net{seed}", einstr, [[sizes[c] for c in t] for t in einstr.split("->")[0].split(",")], "auto", np.float64))
    for seed in range(30):
        einstr, sizes = fz.random_pair_f32_case(np.random.default_rng(5000 + seed))
        cases.append((f"pair32_{seed}", einstr, [[sizes[c] for c in t] for t in einstr.split("->")[0].split(",")], "auto", np.float32))
    for seed in range(24):
        for dtype in ("float32", "float64"):
            einstr, sizes = fz._large_tile_case(np.random.default_rng(9000 + seed), dtype)
            cases.append((f"tile{seed}{dtype}", einstr, [[sizes[c] for c in t] for t in einstr.split("->")[0].split(",")],
                          ((0, 1),), np.dtype(dtype)))
    # the benchmark networks at their real sizes (shapes only: a plan needs no data)
    tn, ssa = nets.mps_overlap(TN, 100, 2, 4, dtype=np.float32, seed=3)
    shapes = [tuple(256 if (d == 2) else d for d in p.shape) for p in tn.params]
    cases.append(("mps100", tn.einsum_str, shapes, ssa_to_linear(ssa, 200), np.float32))
    cases.append(("mps100_f64", tn.einsum_str, shapes, ssa_to_linear(ssa, 200), np.float64))
    tn = nets.peps_closed(TN, 8, 8, 2, dtype=np.float32, seed=6)
    shapes = [tuple(8 if (d == 2 and p.ndim > 1 and ax > 0) else d for ax, d in enumerate(p.shape)) for p in tn.params]
    cases.append(("peps8x8_D8_row", tn.einsum_str, shapes, ssa_to_linear(nets.peps_row_path(8, 8), 128), np.float32))
    cases.append(("cp1024", "ac,ad,ae->cde", [(1024, 1024)] * 3, "auto", np.float32))
    cases.append(("tucker1024", "abc,ae,bf,cg->efg", [(1024,) * 3] + [(1024, 1024)] * 3, "auto", np.float32))
    # tensors of 2^31 elements and more: 64-bit batch offsets, outer free labels moved into the batch group
    cases.append(("wide_gemm", "ak,kb,b->a", [(1 << 17, 16), (16, 1 << 15), (1 << 15,)], ((0, 1), (0, 1)), np.float32))
    cases.append(("wide_outer", "a,b,b->a", [(1 << 16,), (1 << 15,), (1 << 15,)], ((0, 1), (0, 1)), np.float32))
    cases.append(("wide_input", "abc,c->ab", [(1 << 10, 1 << 11, 1 << 10), (1 << 10,)], "auto", np.float64))
    tn, inputs = nets.batched_mps(TN, 20, 64, 4, 512, dtype=np.float32, seed=4)
    ops = E.make_arg_packer(tn)(tn.params, inputs)
    cases.append(("batched_mps", tn.einsum_str, [o.shape for o in ops], ssa_to_linear(nets.batched_mps_path(20), 40), np.float32))
    return cases


@pytest.fixture(scope="module")
def asan_binary():
    proc = subprocess.run(["make", "-C", CSRC, "asan"], capture_output=True, text=True)
    assert proc.returncode == 0, proc.stderr[-3000:]
    assert os.path.exists(BINARY)
    return BINARY


def test_planner_is_clean_under_asan_and_ubsan(asan_binary, tmp_path):
    cases = all_cases()
    texts, flops = zip(*[describe(e, s, o, d) for _n, e, s, o, d in cases])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    proc = subprocess.run([asan_binary], input="\n".join(texts) + "\n", capture_output=True, text=True, env=env, timeout=600)
    assert proc.stderr == "", proc.stderr[-4000:]
    assert proc.returncode == 0, proc.stdout[-2000:]
    lines = proc.stdout.strip().splitlines()
    assert len(lines) == len(cases) >= 250
    for (name, *_), line, fl in zip(cases, lines, flops):
        assert " rc=0 " in line and line.endswith(" ok"), (name, line)
        got = float(line.split("flops=")[1].split()[0])
        assert got == fl, (name, got, fl)          # the sanitizer build plans exactly what the product plans


def test_planner_rejects_bad_descriptions_cleanly_under_asan(asan_binary):
    bad = "\n".join([
        "plan 0 2 1", "in 2 4 5 1 2", "in 2 6 3 2 3", "step 0 1 2 1 3",      # label 2 has extent 5 and 6
        "plan 0 2 1", "in 1 4 1", "in 1 4 1", "step 0 5 1 1",                # operand id out of range
        "plan 1 1 1", "in 2 3 3 1 1", "step 0 -1 1 7",                        # output label in no operand
    ]) + "\n"
    proc = subprocess.run([asan_binary], input=bad, capture_output=True, text=True, timeout=60)
    assert proc.stderr == "" and proc.returncode == 0
    lines = proc.stdout.strip().splitlines()
    assert len(lines) == 3 and all(" rc=-" in ln for ln in lines), lines
