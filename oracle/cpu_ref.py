"""CPU oracle: NumPy restatement of ContracTN's stabilised contraction executor.

TEST INFRASTRUCTURE ONLY.  Nothing under ``contractn_amd/`` imports this file;
it is used by ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` as the checker / reported CPU baseline, never as the thing
that is shipped or measured as the product.

Parity status: PINNED.
  (1) against the reference's own known answers - README.md:33 ([1, 0.36603234]),
      README.md:56-57 (einsum strings), README.md:73-76 ((1,1,1), 1098.61228867 / inf),
      contractn/tests/test_einsum.py:28-64 (all-ones MPS == prod(bond dims)) -
      see tests/test_oracle.py;
  (2) against outputs of the unmodified reference files executed in the build
      container (oracle/gen_golden.py -> tests/golden/*.npz), bit for bit in fp64.

Each function cites the reference lines it follows.  The per-step array math is
NumPy's own ``tensordot`` / ``transpose`` / ``einsum`` - exactly what the
reference reaches through ``opt_einsum``'s NumPy backend.
"""
import operator
from functools import reduce

import numpy as np

MIN_NORM = 1e-7  # reference einsum.py:94

_ASCII = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ"


def stabilize(tensor, log_scale):
    """reference einsum.py:89-107 (NumPy backend): move the mean |T| into the log register."""
    norm = np.sum(np.abs(tensor))
    numel = reduce(operator.mul, tensor.shape, 1)
    rescale = norm / numel
    cond = norm > MIN_NORM
    with np.errstate(divide="ignore", invalid="ignore"):
        tensor = np.where(cond, tensor / rescale, tensor)
        log_scale = np.where(cond, log_scale + np.log(rescale), log_scale)
    return tensor, log_scale


def stabilize_torch(tensor, log_scale):
    """reference einsum.py:89-107 as its TORCH backend evaluates it (einsum.py:9-21, register from
    ``torch.zeros(())`` at :338): every scalar - norm, rescale, log, the register itself - has the tensor's dtype,
    so for float32 tensors the register is accumulated in fp32 (last row).  Restated on NumPy
    arrays of that dtype (a float64 tensor promotes the fp32 zero on the first add: register float64)."""
    dt = tensor.dtype.type
    norm = dt(np.sum(np.abs(tensor)))
    numel = reduce(operator.mul, tensor.shape, 1)
    rescale = dt(norm / dt(numel))
    if norm > MIN_NORM:
        tensor = tensor / rescale
        log_scale = dt(dt(log_scale) + np.log(rescale))
    return tensor, log_scale


def destabilize(tensor, log_scale):
    """reference einsum.py:110-114."""
    return tensor * np.exp(log_scale)


def _einsum_remapped(step_str, *ops):
    """np.einsum with symbols remapped onto a-zA-Z per call (what opt_einsum's
    ``_einsum`` does for the NumPy backend; call site reference einsum.py:382-384)."""
    table = {}
    out = []
    for ch in step_str:
        if ch in ",->":
            out.append(ch)
        else:
            if ch not in table:
                table[ch] = _ASCII[len(table)]
            out.append(table[ch])
    return np.einsum("".join(out), *ops)


def core_contract(operands, contract_list, record=False, torch_register=False):
    """reference einsum.py:326-393: the stabilised pairwise loop.

    ``contract_list[k] = (positions_desc, idx_removed, "L,R->O", _, blas_flag)``.
    Returns ``(last operand, log_scale, rescales)``; with ``record=True`` ``rescales``
    holds every step's rescale factor (0.0 where a step was not rescaled).
    """
    operands = list(operands)
    log_scale = np.float32(0) if torch_register else np.zeros(())  # einsum.py:338 (torch: torch.zeros(()) is fp32)
    rescales = []
    for inds, idx_rm, step_str, _rest, blas_flag in contract_list:
        tmp = [operands.pop(x) for x in inds]  # einsum.py:344
        if blas_flag and "EINSUM" not in blas_flag:  # einsum.py:347
            input_str, results_index = step_str.split("->")
            input_left, input_right = input_str.split(",")
            tensor_result = "".join(s for s in input_left + input_right if s not in idx_rm)
            if idx_rm:
                pairs = sorted((input_left.find(s), input_right.find(s)) for s in idx_rm)
                axes = tuple(zip(*pairs))  # einsum.py:357-365
            else:
                axes = ((), ())
            new_view = np.tensordot(tmp[0], tmp[1], axes=axes)  # einsum.py:371
            if tensor_result != results_index:  # einsum.py:374-377
                new_view = np.transpose(new_view, tuple(map(tensor_result.index, results_index)))
        else:
            new_view = _einsum_remapped(step_str, *tmp)  # einsum.py:382-384
        if record:
            norm = np.sum(np.abs(new_view))
            rescales.append(float(norm / new_view.size) if norm > MIN_NORM else 0.0)
        new_view, log_scale = (stabilize_torch if torch_register else stabilize)(new_view, log_scale)  # einsum.py:387
        operands.append(new_view)  # einsum.py:390
    return operands[0], log_scale, rescales


# ---------------------------------------------------------------------------
# contraction list for an explicit path (what opt_einsum.contract_path returns
# with einsum_call=True; call site reference einsum.py:313-323)
# ---------------------------------------------------------------------------
def contraction_list(einstr, shapes, path):
    lhs, out = einstr.split("->")
    terms = lhs.split(",")
    sizes = {}
    for t, shp in zip(terms, shapes):
        for s, d in zip(t, shp):
            sizes[s] = int(d)
    live = list(terms)
    steps = []
    for num, positions in enumerate(path):
        positions = tuple(sorted(positions, reverse=True))
        picked = [live.pop(p) for p in positions]
        needed = set(out).union(*[set(t) for t in live]) if live else set(out)
        involved = set("".join(picked))
        removed = involved - needed
        if num == len(path) - 1 and not live:
            result = out
        else:
            result = "".join(s for _, s in sorted((sizes[s], s) for s in involved & needed))
        flag = False
        if len(picked) == 2:
            l, r = picked
            sl, sr = set(l), set(r)
            plain = (len(sl) == len(l) and len(sr) == len(r) and (sl & sr) == removed
                     and set(result) == (sl | sr) - (sl & sr))
            flag = "TDOT" if plain else False
        live.append(result)
        steps.append((positions, removed, ",".join(picked) + "->" + result, None, flag))
    assert len(live) == 1
    return steps


def left_to_right_path(n):
    """(0,1) n-1 times: contract the running result (kept at the end) with the next operand.

    After the first step the intermediate sits at the END of the list, so the
    chain order is obtained with positions (0, last)."""
    path = [(0, 1)]
    for k in range(n - 2):
        path.append((0, n - 2 - k))
    return path if n > 1 else [(0,)]


def contract(einstr, *operands, path=None, split_format=False, torch_register=False):
    """reference einsum.py:190-310 for NumPy operands and an explicit path (``torch_register``: the register
    arithmetic of the reference's torch backend, see `stabilize_torch`)."""
    shapes = [np.shape(o) for o in operands]
    if path is None:
        path = left_to_right_path(len(operands))
    clist = contraction_list(einstr, shapes, path)
    result, log_scale, _ = core_contract([np.asarray(o) for o in operands], clist, torch_register=torch_register)
    if split_format:
        return result, log_scale
    with np.errstate(over="ignore"):
        return destabilize(result, log_scale)
