"""``contract()``: stabilised einsum dispatched to the MI355X engine.

Host mirror of reference contractn/einsum.py with the same names, arguments and
error behaviour:

* ``make_einstring``   - reference einsum.py:117-130
* ``make_arg_packer``  - reference einsum.py:133-187
* ``contract``         - reference einsum.py:190-310 (kwargs, TypeError on unknown kwarg)
* ``_contract_path``   - reference einsum.py:313-323 (lru-cached contraction list)
* ``_core_contract``   - reference einsum.py:326-393: here it lowers the contraction
  list to SSA form and hands it to the native plan/executor through the C ABI
  (include/ctn_abi.h).  The pairwise loop, tensordot/transpose/einsum and
  ``stabilize`` all run inside HIP kernels; nothing is computed on the CPU
  except the final log accumulation of the per-step rescale factors (done in
  the reference's order and precision so the log-scale register is
  reproducible bit for bit when the abs-sums are exact) and ``destabilize``.

There is no CPU fallback: without the HIP library or a GPU, ``contract`` raises.
"""
import threading
from collections import OrderedDict
from functools import lru_cache

import numpy as np

from . import engine, paths

MIN_NORM = 1e-7  # reference einsum.py:94
_VALID_EINSUM_KWARGS = ("dtype", "order", "casting")


# ---------------------------------------------------------------------------
# compile step: TN -> einsum string / operand packer
# ---------------------------------------------------------------------------
def make_einstring(tn):
    """Einsum string of a TN: dense/clone/input nodes in insertion order are the
    operand terms (copy nodes are skipped - they only force shared symbols),
    danglers in insertion order give the output."""
    terms, free = [], []
    for node in tn.nodes(as_iter=True, copy_nodes=False, danglers=True):
        if node.dangler:
            free.append(node.symbol)
        else:
            terms.append("".join(node.edge_symbols))
    return ",".join(terms) + "->" + "".join(free)


def make_arg_packer(tn):
    """``(params, inputs) -> operand tuple`` for the TN's einsum string.

    ``params`` are the unique dense tensors in node order; clone nodes reuse
    their base node's slot; ``inputs`` fill the input nodes in node order.
    """
    slot_of_param = {}   # dense node name -> index in params
    operand_src = []     # per operand: ("p", param index) or ("i", input index)
    n_inputs = 0
    for node in tn.nodes(as_iter=True, copy_nodes=False, danglers=False):
        kind = node.node_type
        if kind == "dense":
            assert node.name not in slot_of_param
            slot_of_param[node.name] = len(slot_of_param)
            operand_src.append(("p", slot_of_param[node.name]))
        elif kind == "clone":
            base = node.base_node.name
            assert base in slot_of_param, "clone node precedes its base node"
            operand_src.append(("p", slot_of_param[base]))
        elif kind == "input":
            operand_src.append(("i", n_inputs))
            n_inputs += 1
        else:  # pragma: no cover
            raise AssertionError(f"unexpected node type {kind}")
    n_params = len(slot_of_param)

    def arg_packer(params, inputs):
        assert len(params) == n_params
        assert len(inputs) == n_inputs
        return tuple(params[j] if src == "p" else inputs[j] for src, j in operand_src)

    return arg_packer


# ---------------------------------------------------------------------------
# planner: cached contraction list, cached native plan
# ---------------------------------------------------------------------------
@lru_cache()
def _contract_path(einstr, operand_shapes, **kwargs):
    """Cached contraction list in opt_einsum's ``einsum_call=True`` format."""
    oe = paths.use_opt_einsum()
    if oe is not None:
        _, clist = oe.contract_path(einstr, *operand_shapes, shapes=True, einsum_call=True, **kwargs)
        return tuple(clist)
    return paths.contraction_list(einstr, operand_shapes, **kwargs)


def lower_contraction_list(n_operands, contract_list, shapes=None):
    """opt_einsum-style shrinking positions -> SSA steps with integer labels.

    Returns ``(in_labels, steps)``; ``steps[k] = (lhs_id, rhs_id | -1, out_labels)``,
    inputs are ids ``0..n-1`` and step ``k`` defines id ``n+k``.  The operand popped
    from the higher position is the left one (reference einsum.py:344).

    A step on more than two operands (``optimize=False`` - ONE einsum over everything - or an explicit path with
    n-ary steps; the reference hands any step to ``_einsum``, einsum.py:382-384) is lowered to a chain of pairwise
    steps in the order the greedy finder picks for that sub-network (``shapes`` are needed for that; without them:
    left to right), every intermediate keeping the labels the rest of the step or its result still need.  The
    result ``(T_hat, c)`` is that of the single einsum up to rounding: ``c = ln mean|T|`` whatever the order
    (SURVEY.md App. A); the native plan then simply has more steps than the contraction list.
    """
    live = list(range(n_operands))
    term_of = {}
    # first pass: the term of every network input (needed for the extents before any n-ary step is ordered)
    probe = list(range(n_operands))
    fake = n_operands
    for num, (inds, _idx_rm, step_str, _rest, _flag) in enumerate(contract_list):
        ids = [probe.pop(p) for p in inds]
        parts = step_str.split("->")[0].split(",")
        if len(parts) != len(ids) or not ids:
            raise ValueError(f"step {num}: {len(parts)} terms for {len(ids)} operands")
        for tid, term in zip(ids, parts):
            if tid < n_operands:
                term_of.setdefault(tid, term)
        probe.append(fake)
        fake += 1
    sizes = {}
    if shapes is not None:
        for i, term in term_of.items():
            for c, d in zip(term, shapes[i]):
                sizes[c] = int(d)
    steps = []

    def emit(lhs, rhs, out):
        steps.append((lhs, rhs, tuple(ord(c) for c in out)))
        return n_operands + len(steps) - 1

    for num, (inds, _idx_rm, step_str, _rest, _flag) in enumerate(contract_list):
        ids = [live.pop(p) for p in inds]
        lhs, out = step_str.split("->")
        parts = lhs.split(",")
        if len(ids) <= 2:
            live.append(emit(ids[0], ids[1] if len(ids) == 2 else -1, out))
            continue
        # n-ary step: pairwise chain over (id, term) items
        items = list(zip(ids, parts))
        if sizes and all(c in sizes for t in parts for c in t):
            order = paths.find_path(parts, out, sizes, "greedy")
        else:
            order = [(0, 1)] * (len(items) - 1)
        for k, pos in enumerate(order):
            pos = tuple(sorted(pos, reverse=True))
            if len(pos) != 2:
                raise NotImplementedError(f"step {num}: cannot order a {len(ids)}-operand step pairwise")
            (ia, ta), (ib, tb) = items.pop(pos[0]), items.pop(pos[1])
            if k == len(order) - 1:
                res = out
            else:
                needed = set(out).union(*[set(t) for _i, t in items])
                res = "".join(c for c in dict.fromkeys(ta + tb) if c in needed)
            items.append((emit(ia, ib, res), res))
        if len(items) != 1:
            raise ValueError(f"step {num}: the pairwise order does not reduce its {len(ids)} operands to one")
        live.append(items[0][0])
    if len(live) != 1:
        raise ValueError("contraction list does not reduce the operands to a single tensor")
    missing = [i for i in range(n_operands) if i not in term_of]
    assert not missing, f"operands {missing} never contracted"
    in_labels = [tuple(ord(c) for c in term_of[i]) for i in range(n_operands)]
    return in_labels, steps


@lru_cache(maxsize=256)
def _native_plan_cached(contract_list, shapes, dtype_name, free_output_order=False, in_strides=None):
    in_labels, steps = lower_contraction_list(len(shapes), contract_list, shapes)
    for lab, shp in zip(in_labels, shapes):
        if len(lab) != len(shp):
            raise ValueError(f"operand of shape {shp} does not match its {len(lab)} subscripts")
    return engine.Plan(dtype_name, in_labels, shapes, steps, stabilize=True, min_norm=MIN_NORM,
                       free_output_order=free_output_order, in_strides=in_strides)


_PLAN_LOCK = threading.Lock()


def _native_plan(contract_list, shapes, dtype_name, free_output_order=False, in_strides=None):
    """Cached native plan.  (``lru_cache`` does not serialise misses: without the lock two threads asking for
    the same new plan would each build one, and executors are keyed by plan identity.)  ``in_strides``: element
    strides per operand axis (tuples), None = C-contiguous operands."""
    with _PLAN_LOCK:
        return _native_plan_cached(contract_list, shapes, dtype_name, free_output_order, in_strides)


_native_plan.cache_clear = _native_plan_cached.cache_clear
_native_plan.cache_info = _native_plan_cached.cache_info


# Executors own device memory (workspace, tables, staging).  They are kept in a small LRU so that a
# process contracting many differently-shaped networks does not accumulate workspaces without bound;
# an evicted executor frees its device memory immediately (the plan itself is host-only and cheap).
#
# Threads: the reference's ``contract`` is re-entrant, a native executor is not (one workspace, one pointer
# table, one graph capture).  Executors are therefore cached PER THREAD - the key carries the caller's thread
# id - so two threads contracting the same network run on two executors concurrently instead of queueing on
# one; in addition every use holds the executor's own lock from launch to fetch (an evicting thread waits in
# ``close`` for a run in flight).
_EXECUTOR_LRU = OrderedDict()
_EXECUTOR_LRU_LOCK = threading.Lock()
MAX_CACHED_EXECUTORS = 16


def _executor_for(plan, replicas=1, device=0, stream=None):
    key = (id(plan), replicas, device, stream, threading.get_ident())
    evicted = []
    with _EXECUTOR_LRU_LOCK:
        ex = _EXECUTOR_LRU.get(key)
        if ex is not None:
            _EXECUTOR_LRU.move_to_end(key)
            return ex
    ex = engine.Executor(plan, replicas=replicas, device=device, stream=stream)
    with _EXECUTOR_LRU_LOCK:
        _EXECUTOR_LRU[key] = ex
        while len(_EXECUTOR_LRU) > MAX_CACHED_EXECUTORS:
            evicted.append(_EXECUTOR_LRU.popitem(last=False)[1])
    for old in evicted:   # outside the table lock: close() waits for a run that is still in flight
        old.close()
    return ex


class _locked_executor:
    """``with _locked_executor(plan, ...) as ex``: the cached executor of this thread, locked and ALIVE.  Eviction is
    global while the cache is per thread: another thread's insertion can evict - and close - this thread's executor
    between the lookup and the lock.  A closed executor (handle gone) is simply looked up again."""

    def __init__(self, plan, replicas=1, device=0, stream=None):
        self.args = (plan, replicas, device, stream)
        self.ex = None

    def __enter__(self):
        while True:
            ex = _executor_for(*self.args)
            ex.lock.acquire()
            if getattr(ex, "is_open", lambda: True)():
                self.ex = ex
                return ex
            ex.lock.release()          # closed under our feet: it is out of the table already

    def __exit__(self, *exc):
        self.ex.lock.release()
        return False


def clear_caches():
    """Drop every cached executor (device memory), native plan and contraction path."""
    with _EXECUTOR_LRU_LOCK:
        dropped = list(_EXECUTOR_LRU.values())
        _EXECUTOR_LRU.clear()
    for ex in dropped:
        ex.close()
    _native_plan.cache_clear()
    _contract_path.cache_clear()


# ---------------------------------------------------------------------------
# scale register (host part of stabilize / destabilize)
# ---------------------------------------------------------------------------
def accumulate_log_scale(step_rescales, dtype, register_dtype=np.float64):
    """Sum of log(rescale) over the steps, in path order, with the reference's
    NumPy semantics (reference einsum.py:104-106): ``log`` is
    evaluated in the tensor dtype, the register is a float64 0-d array.

    ``register_dtype``: the reference's TORCH backend keeps the register in the tensor dtype - a
    ``torch.float32`` 0-d tensor (``torch.zeros(())``, einsum.py:338) that every step adds to in fp32 - so torch
    operands pass their dtype here: same logs, the strictly sequential adds in that precision."""
    resc = np.asarray(step_rescales, dtype=np.float64)
    reg = np.dtype(register_dtype)
    logs = np.zeros(resc.shape, dtype=reg)
    mask = resc > 0
    logs[mask] = np.log(resc[mask].astype(dtype)).astype(reg)
    if logs.size == 0:
        return np.zeros((), dtype=reg)
    return np.asarray(np.add.accumulate(logs)[-1])  # strictly sequential adds, in the register's precision


def stabilize(tensor, log_scale, backend="auto"):
    """Move the mean magnitude of ``tensor`` into the ``log_scale`` register and return both
    (reference einsum.py:89-107): ``(tensor / s, log_scale + log s)`` with ``s = sum|tensor| / numel``
    when ``sum|tensor| > 1e-7``, otherwise both unchanged.

    Runs on the device as a one-operand plan (the same abs-sum / rescale / finalize kernels every
    pairwise step ends with), so like ``contract`` it raises without the HIP library or a GPU.
    """
    ndim = len(tensor.shape)
    if ndim > 52:
        raise ValueError("stabilize: more than 52 axes")
    sub = "".join(chr(ord("a") + i) if i < 26 else chr(ord("A") + i - 26) for i in range(ndim))
    t_hat, c = contract(sub + "->" + sub, tensor, split_format=True, backend=backend)
    return t_hat, log_scale + c


def destabilize(tensor, log_scale, backend="numpy"):
    """``tensor * exp(log_scale)`` (reference einsum.py:110-114); may overflow to inf by design."""
    if backend == "torch":
        import torch

        return tensor * torch.exp(log_scale)
    return tensor * np.exp(log_scale)


# ---------------------------------------------------------------------------
# executor seam
# ---------------------------------------------------------------------------
def parse_backend(arrays, backend="auto"):
    """Name of the array library of the operands ('numpy' or 'torch')."""
    if backend not in (None, "auto"):
        return backend
    for a in arrays:
        mod = type(a).__module__.split(".")[0]
        if mod == "torch":
            return "torch"
    return "numpy"


def _common_dtype(operands, backend, requested=None):
    """The engine's arithmetic type for these operands: float32 or float64 (ints and bools compute in
    float64); complex operands compute in the real type of their components (`_core_contract_complex`)."""
    if requested is not None:
        dt = np.dtype(requested)
    elif backend == "torch":
        import torch

        wide = (torch.float64, torch.complex128)
        dt = np.dtype(np.float64 if any(o.dtype in wide for o in operands) else np.float32)
    else:
        dt = np.result_type(*[np.asarray(o).dtype for o in operands])
    if dt.kind == "c":
        dt = np.dtype(np.float32 if dt == np.complex64 else np.float64)
    if dt == np.float32:
        return np.dtype(np.float32)
    if dt.kind == "f" and dt.itemsize < 4:
        return np.dtype(np.float32)
    return np.dtype(np.float64)  # ints, bools, float64 -> float64


def _core_contract(operands, contract_list, backend="numpy", _plain=False, **einsum_kwargs):
    """Run a contraction list on the GPU; returns ``(rescaled result, log_scale)``.

    Same seam as reference einsum.py:326-393: ``contract_list`` is the list of
    5-tuples produced by ``contract_path(..., einsum_call=True)``.

    ``_plain`` (this module's `contract(split_format=False)` only): where the result stays on the device the
    de-stabilised tensor ``T_hat * exp(log_scale)`` (einsum.py:110-114) is formed in the same pass as the last division
    by the rescale factor, and ``(tensor, None)`` is returned; everywhere else the flag is ignored.
    """
    operands = list(operands)
    contract_list = tuple(
        (tuple(c[0]), frozenset(c[1]), c[2], None, c[4]) for c in contract_list
    )
    dtype = _common_dtype(operands, backend, einsum_kwargs.get("dtype"))
    shapes = tuple(tuple(int(d) for d in op.shape) for op in operands)
    if _is_complex(operands, backend):
        return _core_contract_complex(operands, contract_list, shapes, dtype, backend)
    plan = _native_plan(contract_list, shapes, dtype.name)
    if backend == "torch":
        return _run_torch(plan, operands, dtype, plain=_plain)
    with _locked_executor(plan, 1) as ex:
        outs, _dev_log, resc = ex.run_host([operands])
    log_scale = accumulate_log_scale(resc[0], dtype)
    return outs[0], log_scale


# ---------------------------------------------------------------------------
# complex tensors: the same engine on their (re, im) components
# ---------------------------------------------------------------------------
def _is_complex(operands, backend):
    if backend == "torch":
        return any(o.dtype.is_complex for o in operands)
    return any(np.asarray(o).dtype.kind == "c" for o in operands)


_COMPLEX_LABEL = 1 << 24          # integer labels of the (re, im) legs start here (symbols are code points < 2^21)
# z = x * y on components: z_c = sum_ab S[a, b, c] x_a y_b
_CSTRUCT = np.zeros((2, 2, 2))
_CSTRUCT[0, 0, 0], _CSTRUCT[1, 1, 0], _CSTRUCT[0, 1, 1], _CSTRUCT[1, 0, 1] = 1.0, -1.0, 1.0, 1.0


@lru_cache(maxsize=64)
def _complex_plan_cached(contract_list, shapes, is_cplx, dtype_name):
    """The native plan of a network with complex operands, on REAL tensors: a complex operand is its array of
    (re, im) pairs - one more, innermost axis of extent 2 with a label of its own - and a pairwise step on two complex
    tensors ``z = x y`` is ``z_c = sum_ab S_abc x_a y_b`` with the 2 x 2 x 2 structure tensor S of complex
    multiplication: S is contracted into the smaller operand (a streaming step that doubles it: the 2 x 2 real matrix
    of every element), then ONE real GEMM with the pair leg as one more contracted label does the step.  A real operand
    meeting a complex one needs nothing (the pair leg rides along as a free label).  The reference reaches complex
    arithmetic through NumPy (einsum.py:371-384; SURVEY.md App. A row complex128); here it is four real multiply-adds per
    complex one on the same kernels.  Returns ``(plan, S inputs appended, result is complex)``."""
    n = len(shapes)
    in_labels, steps = lower_contraction_list(n, contract_list, shapes)
    sizes = {}
    for lab, shp in zip(in_labels, shapes):
        for l_, d_ in zip(lab, shp):
            sizes[l_] = int(d_)
    fresh = [_COMPLEX_LABEL]

    def new_label():
        fresh[0] += 1
        return fresh[0]

    cplx = {}                                   # SSA id -> label of its (re, im) leg
    labels_of = {}
    new_in_labels, new_shapes = [], []
    for i in range(n):
        lab = tuple(in_labels[i])
        shp = tuple(shapes[i])
        if is_cplx[i]:
            cplx[i] = new_label()
            lab, shp = lab + (cplx[i],), shp + (2,)
        labels_of[i] = lab
        new_in_labels.append(lab)
        new_shapes.append(shp)
    # first pass: how many S operands are needed (they are appended to the inputs, so ids shift by their count)
    n_s = 0
    probe = dict(cplx)
    for k, (lhs, rhs, _out) in enumerate(steps):
        both = rhs >= 0 and lhs in probe and rhs in probe
        n_s += 1 if both else 0
        if lhs in probe or (rhs >= 0 and rhs in probe):
            probe[n + k] = True
    n_in = n + n_s
    remap = {i: i for i in range(n)}             # old SSA id -> new SSA id
    new_steps, s_used = [], 0

    def numel(lab):
        v = 1
        for l_ in lab:
            v *= sizes.get(l_, 2)
        return v

    for k, (lhs, rhs, out) in enumerate(steps):
        out = tuple(out)
        a, b = remap[lhs], (remap[rhs] if rhs >= 0 else -1)
        ca, cb = cplx.get(lhs), (cplx.get(rhs) if rhs >= 0 else None)
        if ca is not None and cb is not None:
            xo = new_label()
            s_id = n + s_used
            s_used += 1
            small_is_lhs = numel(labels_of[lhs]) <= numel(labels_of[rhs])
            x_small, x_big = (ca, cb) if small_is_lhs else (cb, ca)
            small, big = (a, b) if small_is_lhs else (b, a)
            small_lab = labels_of[lhs] if small_is_lhs else labels_of[rhs]
            new_in_labels.append((ca, cb, xo))
            new_shapes.append((2, 2, 2))
            widened = tuple(l_ for l_ in small_lab if l_ != x_small) + (x_big, xo)
            new_steps.append((small, s_id, widened))                     # the element's 2 x 2 real matrix
            mid = n_in + len(new_steps) - 1
            first, second = (mid, big) if small_is_lhs else (big, mid)   # keep the step's left / right operands
            new_steps.append((first, second, out + (xo,)))
            cplx[n + k] = xo
        elif ca is not None or cb is not None:
            x = ca if ca is not None else cb
            new_steps.append((a, b, out + (x,)))
            cplx[n + k] = x
        else:
            new_steps.append((a, b, out))
        labels_of[n + k] = out + ((cplx[n + k],) if n + k in cplx else ())
        remap[n + k] = n_in + len(new_steps) - 1
    plan = engine.Plan(dtype_name, new_in_labels, new_shapes, new_steps, stabilize=True, min_norm=MIN_NORM)
    return plan, n_s, (n + len(steps) - 1) in cplx


def _core_contract_complex(operands, contract_list, shapes, dtype, backend):
    """`_core_contract` for networks with complex operands (see `_complex_plan_cached`).  The engine normalises by the
    mean of |re| + |im|; the reference by the mean modulus (einsum.py:97 ``abs``): the final tensor is brought to the
    reference's normalisation here - one pass over the result - and the register moves by the log of the ratio."""
    if backend == "torch":
        import torch

        is_c = tuple(bool(o.dtype.is_complex) for o in operands)
    else:
        operands = [np.asarray(o) for o in operands]
        is_c = tuple(o.dtype.kind == "c" for o in operands)
    with _PLAN_LOCK:
        plan, n_s, out_complex = _complex_plan_cached(contract_list, shapes, is_c, dtype.name)
    cdt = np.dtype(np.complex64 if dtype == np.float32 else np.complex128)
    if backend == "torch":
        tdt = torch.float32 if dtype == np.float32 else torch.float64
        ctd = torch.complex64 if dtype == np.float32 else torch.complex128
        dev = next((o.device for o in operands if o.is_cuda), torch.device("cpu"))
        host = []
        for o, c in zip(operands, is_c):
            o = o.detach()
            o = torch.view_as_real(o.to(ctd).contiguous()) if c else o.to(tdt)
            host.append(o.cpu().numpy())
    else:
        host = []
        for o, c in zip(operands, is_c):
            if c:
                o = np.ascontiguousarray(o, dtype=cdt)
                host.append(o.view(dtype).reshape(o.shape + (2,)))
            else:
                host.append(np.ascontiguousarray(o, dtype=dtype))
    host += [_CSTRUCT.astype(dtype)] * n_s
    with _locked_executor(plan, 1) as ex:
        outs, _dev_log, resc = ex.run_host([host])
    log_scale = accumulate_log_scale(resc[0], dtype, register_dtype=dtype if backend == "torch" else np.float64)
    res = outs[0]
    if out_complex:
        res = np.ascontiguousarray(res).view(cdt).reshape(res.shape[:-1])
        norm = np.sum(np.abs(res))               # the reference's norm: moduli (einsum.py:97)
        if norm > MIN_NORM:
            ratio = (norm / res.size).astype(dtype) if hasattr(norm, "astype") else dtype.type(norm / res.size)
            res = (res / ratio).astype(cdt)
            log_scale = np.asarray(log_scale + np.log(ratio).astype(log_scale.dtype), dtype=log_scale.dtype)
    if backend == "torch":
        return torch.from_numpy(np.array(res)).to(dev), torch.tensor(float(log_scale), dtype=tdt, device=dev)
    return res, log_scale


def _run_torch(plan, operands, dtype, plain=False):
    import torch

    tdt = torch.float32 if dtype == np.float32 else torch.float64
    if torch.is_grad_enabled() and any(o.requires_grad for o in operands):
        # the reference's torch backend is differentiable (einsum.py:9-21); this engine writes into a fresh
        # buffer outside autograd, so a training loop would silently get no gradients - refuse instead
        raise NotImplementedError(
            "the HIP engine does not build an autograd graph: detach() the operands or contract under "
            "torch.no_grad()")
    if not all(o.is_cuda for o in operands):
        host = [o.detach().cpu().numpy() for o in operands]
        with _locked_executor(plan, 1) as ex:
            outs, _dev_log, resc = ex.run_host([host])
        log_scale = accumulate_log_scale(resc[0], dtype, register_dtype=dtype)
        return torch.from_numpy(np.array(outs[0])), torch.tensor(float(log_scale), dtype=tdt)   # (a closed network: 0-d)
    dev = operands[0].device
    ops = [o.to(device=dev, dtype=tdt).contiguous() for o in operands]
    # views into a larger storage may start at an odd offset: vector loads need 16-byte alignment
    ops = [o.clone() if o.data_ptr() % 16 else o for o in ops]
    out = torch.empty(plan.out_shape, dtype=tdt, device=dev)
    tstream = torch.cuda.current_stream(dev)
    stream = tstream.cuda_stream
    if not stream:
        # torch is on the legacy default stream (handle 0), which cannot be handed to the executor:
        # it runs on its own non-blocking stream, so wait for the producers of the operands first
        tstream.synchronize()
    with _locked_executor(plan, 1, device=dev.index or 0, stream=stream) as ex:
        if plain:
            ex.set_finish_mode(1)
        try:
            ex.enqueue([o.data_ptr() for o in ops], [out.data_ptr()])
            _dev_log, resc = ex.fetch()
            # the torch backend's register: the tensor dtype, sequential adds in it (reference einsum.py:338; App. A)
            log_scale = accumulate_log_scale(resc[0], dtype, register_dtype=dtype)
            if plain:
                # destabilize (einsum.py:110-114) in the pass that divides by the last rescale factor: the factor is
                # exp(register) evaluated in the register's dtype, like `tensor * torch.exp(log_scale)`
                factor = torch.exp(torch.tensor(float(log_scale), dtype=tdt))
                ex.finish([float(factor)])
        finally:
            if plain:
                ex.set_finish_mode(0)
    if plain:
        return out, None
    return out, torch.tensor(float(log_scale), dtype=tdt, device=dev)


# ---------------------------------------------------------------------------
# public entry point
# ---------------------------------------------------------------------------
def contract(*operands, **kwargs):
    """Stabilised einsum: ``contract(subscripts, *operands, split_format=False,
    optimize=True, use_blas=True, memory_limit=None, backend="auto", dtype=None,
    order="K", casting="safe")``.

    Drop-in for reference ``contractn.contract`` (einsum.py:190-310).  With
    ``split_format=True`` returns ``(rescaled_output, log_scale)`` where
    ``output == rescaled_output * exp(log_scale)``; otherwise the de-stabilised
    product (which may overflow, as in the reference).
    """
    optimize_arg = kwargs.pop("optimize", True)
    if optimize_arg is True:
        optimize_arg = "auto"
    elif not isinstance(optimize_arg, (str, bool)) and optimize_arg is not None:
        optimize_arg = tuple(tuple(int(p) for p in step) for step in optimize_arg)  # hashable

    einsum_kwargs = {k: v for k, v in kwargs.items() if k in _VALID_EINSUM_KWARGS}
    use_blas = kwargs.pop("use_blas", True)
    split_format = kwargs.pop("split_format", False)
    memory_limit = kwargs.pop("memory_limit", None)
    backend = kwargs.pop("backend", "auto")

    unknown = [k for k in kwargs if k not in _VALID_EINSUM_KWARGS]
    if unknown:
        raise TypeError("Did not understand the following kwargs: {}".format(unknown))

    assert isinstance(operands[0], str)
    einstr, tensors = operands[0], operands[1:]
    backend = parse_backend(tensors, backend)
    if backend not in ("numpy", "torch"):
        raise NotImplementedError(
            f"backend '{backend}' is not supported: the MI355X engine takes numpy or torch tensors"
        )
    shapes = tuple(tuple(int(d) for d in op.shape) for op in tensors)
    contract_list = _contract_path(
        einstr, shapes, optimize=optimize_arg, memory_limit=memory_limit, use_blas=use_blas
    )
    result, log_scale = _core_contract(tensors, contract_list, backend, _plain=not split_format, **einsum_kwargs)
    if split_format:
        return result, log_scale
    if log_scale is None:          # (a device-resident result: already de-stabilised, in the pass of its last division)
        return result
    return destabilize(result, log_scale, backend)


# ---------------------------------------------------------------------------
# throughput API: R independent contractions of one network per launch sequence
# ---------------------------------------------------------------------------
class BatchedContraction:
    """``replicas`` independent contractions of one compiled network in flight at once.

    A single contraction of a dependent chain cannot fill 256 CUs;
    this is the grouped form: every pairwise step is ONE kernel launch covering
    all replicas.  Operands are device pointers (``enqueue``) or NumPy arrays
    (``run_host``).  Results: ``(t_hat [R, ...], log_scale [R])`` in split format.
    """

    def __init__(self, einstr, shapes, dtype, optimize="auto", replicas=1, device=0, stream=None,
                 memory_limit=None, use_blas=True, free_output_order=False, in_strides=None):
        if not isinstance(optimize, (str, bool)) and optimize is not None:
            optimize = tuple(tuple(int(p) for p in step) for step in optimize)
        shapes = tuple(tuple(int(d) for d in s) for s in shapes)
        self.einsum_str = einstr
        self.dtype = np.dtype(dtype)
        self.contract_list = _contract_path(einstr, shapes, optimize=optimize,
                                            memory_limit=memory_limit, use_blas=use_blas)
        # free_output_order: the result's axis order is the engine's choice (``out_subscripts`` tells which)
        if in_strides is not None:
            in_strides = tuple(tuple(int(x) for x in st_) for st_ in in_strides)
        self.plan = _native_plan(self.contract_list, shapes, self.dtype.name, free_output_order, in_strides)
        self.out_subscripts = "".join(chr(l_) for l_ in self.plan.out_labels)
        self.replicas = int(replicas)
        self.executor = engine.Executor(self.plan, replicas=self.replicas, device=device, stream=stream)

    def run_host(self, operand_sets):
        outs, _dev_log, resc = self.executor.run_host(operand_sets)
        logs = np.array([accumulate_log_scale(resc[r], self.dtype) for r in range(self.replicas)])
        return outs, logs

    def enqueue(self, in_ptrs, out_ptrs):
        self.executor.enqueue(in_ptrs, out_ptrs)

    def fetch_log_scale(self):
        """Wait for the last enqueue; log-scale registers accumulated in the reference's order."""
        _dev_log, resc = self.executor.fetch()
        return np.array([accumulate_log_scale(resc[r], self.dtype) for r in range(self.replicas)])
