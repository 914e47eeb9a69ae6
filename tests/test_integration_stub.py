"""The ctypes stub printed in INTEGRATION.md is executable: extract it and run it against the library."""
import os
import re

import numpy as np
import pytest

from contractn_amd import engine, paths
from tests.helpers import ROOT, load_golden


def stub_namespace():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.search(r"```python\n# contractn/hip_backend.py.*?\n(.*?)```", text, flags=re.S).group(1)
    code = code.replace('C.CDLL("libctn_hip.so")', f'C.CDLL("{engine.LIB_PATH}")')
    ns = {}
    exec(compile(code, "INTEGRATION.md:stub", "exec"), ns)
    return ns


def test_stub_parses_and_lowers():
    ns = stub_namespace()
    g = load_golden("mps_overlap_6x8x3_f32")
    clist = paths.contraction_list(g["einsum_str"], [o.shape for o in g["operands"]], optimize=g["path"])
    labels, steps = ns["_lower"](len(g["operands"]), clist)
    assert len(labels) == 12 and len(steps) == 11 and steps[-1][2] == []


@pytest.mark.gpu
def test_stub_runs_on_gpu():
    ns = stub_namespace()
    g = load_golden("mps_overlap_5x64x4_f32")
    clist = paths.contraction_list(g["einsum_str"], [o.shape for o in g["operands"]], optimize=g["path"])
    out, log_scale = ns["core_contract_hip"](g["operands"], clist)
    assert abs(float(out) - float(g["t_hat"])) <= 1e-4
    assert abs(float(log_scale) - float(g["log_scale"])) <= 1e-4
