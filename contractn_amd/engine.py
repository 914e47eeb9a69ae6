"""ctypes binding of ``libctn_hip.so`` (include/ctn_abi.h) - the only way compute is reached.

There is no CPU fallback: if the library is missing, or no gfx950 device is
visible when a contraction is executed, a RuntimeError is raised.  Plans are
pure host objects, so they can be built and inspected without a GPU.
"""
import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CTN_LIB_PATH: development override (A/B of kernel experiment builds, `make EXP=n`)
LIB_PATH = os.environ.get("CTN_LIB_PATH") or os.path.join(_HERE, "lib", "libctn_hip.so")

CTN_F32, CTN_F64 = 0, 1
CTN_MEM_HOST, CTN_MEM_DEVICE = 0, 1
KERNEL_NAMES = {0: "element", 1: "dot", 2: "mfma_f32", 3: "mfma_f64", 4: "rowdot", 5: "fused"}

_STATUS_EXC = {
    -1: AssertionError,
    -2: ValueError,
    -3: NotImplementedError,
    -4: MemoryError,
    -5: RuntimeError,
    -6: RuntimeError,
    -7: RuntimeError,
}

# every symbol include/ctn_abi.h declares
ABI_SYMBOLS = (
    "ctn_version", "ctn_last_error", "ctn_device_count",
    "ctn_plan_create", "ctn_plan_destroy", "ctn_plan_dtype", "ctn_plan_n_inputs",
    "ctn_plan_n_steps", "ctn_plan_flops", "ctn_plan_bytes_min", "ctn_plan_out_ndim",
    "ctn_plan_out_dims", "ctn_plan_out_labels", "ctn_plan_out_numel", "ctn_plan_out_bytes",
    "ctn_plan_workspace_bytes", "ctn_plan_step_info",
    "ctn_exec_create", "ctn_exec_destroy", "ctn_exec_run", "ctn_exec_enqueue",
    "ctn_exec_fetch", "ctn_exec_synchronize", "ctn_exec_set_timing", "ctn_exec_step_ms",
    "ctn_exec_step_tile", "ctn_exec_set_rescale_mode", "ctn_exec_eager_reruns",
    "ctn_exec_snapshot_scales", "ctn_exec_scales_suspect", "ctn_exec_combine_split",
    "ctn_exec_add_scales", "ctn_exec_merge_scales", "ctn_exec_report_suspect",
    "ctn_exec_set_finish_mode", "ctn_exec_finish",
)


class PlanDesc(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32),
        ("n_inputs", C.c_int32),
        ("in_ndim", C.POINTER(C.c_int32)),
        ("in_dims", C.POINTER(C.c_int64)),
        ("in_labels", C.POINTER(C.c_int32)),
        ("in_strides", C.POINTER(C.c_int64)),
        ("n_steps", C.c_int32),
        ("step_lhs", C.POINTER(C.c_int32)),
        ("step_rhs", C.POINTER(C.c_int32)),
        ("step_out_ndim", C.POINTER(C.c_int32)),
        ("step_out_labels", C.POINTER(C.c_int32)),
        ("stabilize", C.c_int32),
        ("min_norm", C.c_double),
    ]


class StepInfo(C.Structure):
    _fields_ = [
        ("kernel", C.c_int32),
        ("swapped", C.c_int32),
        ("batch", C.c_int64),
        ("m", C.c_int64),
        ("n", C.c_int64),
        ("k", C.c_int64),
        ("mode_a", C.c_int32),
        ("mode_b", C.c_int32),
        ("partials", C.c_int32),
        ("blocks", C.c_int32),
        ("flops", C.c_double),
        ("out_numel", C.c_int64),
        ("tile_m", C.c_int32),
        ("tile_n", C.c_int32),
        ("epilogue_sum", C.c_int32),
        ("reserved", C.c_int32),
    ]


_lib = None


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.

    PyTorch-ROCm ships its own ``libamdhip64.so`` (soname ``libamdhip64.so.7``) and asks for it by
    the unversioned file name, so if our library pulled in ``/opt/rocm/lib/libamdhip64.so.7`` first
    a later ``import torch`` would load a second copy of the runtime and see no GPU.  When torch is
    installed, map its copy first (without importing torch): our ``NEEDED libamdhip64.so.7`` then
    resolves to it by soname, whichever of the two is used first.
    """
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load_library():
    """Load the engine library (built by ``__graft_entry__.build()`` / csrc/Makefile)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"HIP engine library not found at {LIB_PATH}; build it with "
            "`make -C contractn_amd/csrc` (there is no CPU fallback)"
        )
    _share_hip_runtime_with_torch()
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, f64 = C.c_void_p, C.c_int, C.c_int64, C.c_double
    sig = {
        "ctn_version": (i32, []),
        "ctn_last_error": (C.c_char_p, []),
        "ctn_device_count": (i32, [C.POINTER(C.c_int)]),
        "ctn_plan_create": (i32, [C.POINTER(PlanDesc), C.POINTER(vp)]),
        "ctn_plan_destroy": (None, [vp]),
        "ctn_plan_dtype": (i32, [vp]),
        "ctn_plan_n_inputs": (i32, [vp]),
        "ctn_plan_n_steps": (i32, [vp]),
        "ctn_plan_flops": (f64, [vp]),
        "ctn_plan_bytes_min": (i64, [vp]),
        "ctn_plan_out_ndim": (i32, [vp]),
        "ctn_plan_out_dims": (i32, [vp, C.POINTER(C.c_int64)]),
        "ctn_plan_out_labels": (i32, [vp, C.POINTER(C.c_int32)]),
        "ctn_plan_out_numel": (i64, [vp]),
        "ctn_plan_out_bytes": (i64, [vp]),
        "ctn_plan_workspace_bytes": (i64, [vp, i32]),
        "ctn_plan_step_info": (i32, [vp, i32, C.POINTER(StepInfo)]),
        "ctn_exec_create": (i32, [vp, i32, vp, i32, C.POINTER(vp)]),
        "ctn_exec_destroy": (None, [vp]),
        "ctn_exec_run": (i32, [vp, C.POINTER(vp), i32, C.POINTER(vp), i32,
                               C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "ctn_exec_enqueue": (i32, [vp, C.POINTER(vp), C.POINTER(vp)]),
        "ctn_exec_fetch": (i32, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "ctn_exec_synchronize": (i32, [vp]),
        "ctn_exec_set_timing": (i32, [vp, i32]),
        "ctn_exec_step_ms": (i32, [vp, C.POINTER(C.c_float)]),
        "ctn_exec_step_tile": (i32, [vp, i32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
        "ctn_exec_set_rescale_mode": (i32, [vp, i32]),
        "ctn_exec_eager_reruns": (i32, [vp]),
        "ctn_exec_snapshot_scales": (i32, [vp, vp, i32, vp]),
        "ctn_exec_scales_suspect": (i32, [vp, vp, i32]),
        "ctn_exec_combine_split": (i32, [vp, i32, vp, i64, vp, i64, i32, i64, vp]),
        "ctn_exec_report_suspect": (i32, [vp, i32]),
        "ctn_exec_set_finish_mode": (i32, [vp, i32]),
        "ctn_exec_finish": (i32, [vp, C.POINTER(C.c_double)]),
        "ctn_exec_add_scales": (i32, [vp, vp, vp, i32, i32, C.POINTER(vp), C.POINTER(vp)]),
        "ctn_exec_merge_scales": (i32, [vp, i32, vp, i64, i64, vp, i32, i32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def _check(rc):
    if rc < 0:
        msg = load_library().ctn_last_error().decode("utf-8", "replace")
        raise _STATUS_EXC.get(rc, RuntimeError)(f"ctn engine error {rc}: {msg}")
    return rc


def device_count():
    """Number of visible HIP devices (0 when there is none)."""
    n = C.c_int(0)
    rc = load_library().ctn_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def _i32(seq):
    return np.ascontiguousarray(np.asarray(list(seq), dtype=np.int32))


def _i64(seq):
    return np.ascontiguousarray(np.asarray(list(seq), dtype=np.int64))


def _ptr(arr, ctype):
    return arr.ctypes.data_as(C.POINTER(ctype))


HOIST_MAX_NUMEL = 1 << 16


def hoist_leaf_steps(n_inputs, in_labels, in_dims, steps, max_numel=HOIST_MAX_NUMEL):
    """Execution order of a path: small steps whose two operands are NETWORK INPUTS (physical legs absorbed into
    their sites: the leaves of a PEPS / MPS contraction tree) move to the front, everything else keeps its order.
    They are independent of each other and of all other steps, and consecutive the engine sends them out as ONE
    launch (`k_stream_group`) instead of dozens of 5-us ones.  Returns ``(steps in execution order, order)`` with
    ``order[new] = old``, or ``(steps, None)`` when nothing moves.  Reporting (step info, rescales - hence the order
    in which the log-scale register is accumulated, which stays the path's - timings, tiles) keeps the caller's step
    numbers; only the order of launches changes.  ``CTN_HOIST=0`` disables."""
    if os.environ.get("CTN_HOIST", "1") == "0" or len(steps) < 3:
        return steps, None
    ext = {}
    for lab, dims in zip(in_labels, in_dims):
        for l, d in zip(lab, dims):
            ext[l] = int(d)

    def numel(labels):
        n = 1
        for l in labels:
            n *= ext.get(l, 1)
        return n

    small_in = [numel(lab) <= max_numel for lab in in_labels]
    leaf = [i for i, (l, r, out) in enumerate(steps[:-1])
            if 0 <= l < n_inputs and 0 <= r < n_inputs and small_in[l] and small_in[r] and numel(out) <= max_numel]
    if len(leaf) < 2 or leaf == list(range(len(leaf))):
        return steps, None
    taken = set(leaf)
    order = leaf + [i for i in range(len(steps)) if i not in taken]
    new_id = {n_inputs + old: n_inputs + new for new, old in enumerate(order)}

    def remap(t):
        return t if t < n_inputs else new_id[t]

    return [(remap(steps[o][0]), remap(steps[o][1]) if steps[o][1] >= 0 else -1, steps[o][2]) for o in order], order


class Plan:
    """Immutable native plan for one (einsum string, shapes, path, dtype)."""

    def __init__(self, dtype, in_labels, in_dims, steps, stabilize=True, min_norm=1e-7,
                 in_strides=None, free_output_order=False):
        """``in_labels``/``in_dims``: per-operand int lists; ``steps``: (lhs, rhs|-1, out_labels).
        ``free_output_order``: the engine also picks the axis order of the final result (``out_labels`` tells which):
        for results that only feed another plan (the stages of a sliced contraction)."""
        lib = load_library()
        self.np_dtype = np.dtype(dtype)
        if self.np_dtype == np.float32:
            code = CTN_F32
        elif self.np_dtype == np.float64:
            code = CTN_F64
        else:
            raise TypeError(f"HIP engine supports float32/float64, got {self.np_dtype}")
        steps = [(int(s[0]), int(s[1]), tuple(s[2])) for s in steps]
        steps, order = hoist_leaf_steps(len(in_labels), in_labels, in_dims, steps)
        # native step index of the caller's step i, and back (identity unless leaf steps were hoisted)
        self._native_of = None if order is None else np.argsort(np.asarray(order)).astype(np.int64)
        self._caller_of = None if order is None else np.asarray(order, dtype=np.int64)
        ndim = _i32(len(lab) for lab in in_labels)
        dims = _i64(d for shp in in_dims for d in shp)
        labels = _i32(l for lab in in_labels for l in lab)
        lhs = _i32(s[0] for s in steps)
        rhs = _i32(s[1] for s in steps)
        ond = _i32(len(s[2]) for s in steps)
        olab = _i32(l for s in steps for l in s[2])
        desc = PlanDesc()
        desc.dtype = code
        desc.n_inputs = len(in_labels)
        desc.in_ndim = _ptr(ndim, C.c_int32)
        desc.in_dims = _ptr(dims, C.c_int64)
        desc.in_labels = _ptr(labels, C.c_int32)
        if in_strides is not None:
            strides = _i64(s for st in in_strides for s in st)
            desc.in_strides = _ptr(strides, C.c_int64)
        desc.n_steps = len(steps)
        desc.step_lhs = _ptr(lhs, C.c_int32)
        desc.step_rhs = _ptr(rhs, C.c_int32)
        desc.step_out_ndim = _ptr(ond, C.c_int32)
        desc.step_out_labels = _ptr(olab, C.c_int32)
        desc.stabilize = (1 if stabilize else 0) | (2 if free_output_order else 0)
        desc.min_norm = float(min_norm)
        handle = C.c_void_p()
        _check(lib.ctn_plan_create(C.byref(desc), C.byref(handle)))
        self._h = handle
        self._lib = lib
        self.n_inputs = len(in_labels)
        self.n_steps = len(steps)
        # element offsets of the operands inside one packed host buffer (Executor.run_host packs many small operands)
        numels = [int(np.prod(shp, dtype=np.int64)) if len(shp) else 1 for shp in in_dims]
        self.in_numel_total = int(sum(numels))
        self.in_offsets_bytes = (np.concatenate([[0], np.cumsum(numels[:-1])]) if numels else np.zeros(0)).astype(np.uint64) \
            * np.uint64(self.np_dtype.itemsize)
        nd = lib.ctn_plan_out_ndim(handle)
        od = (C.c_int64 * max(nd, 1))()
        _check(lib.ctn_plan_out_dims(handle, od))
        self.out_shape = tuple(int(od[i]) for i in range(nd))
        ol = (C.c_int32 * max(nd, 1))()
        _check(lib.ctn_plan_out_labels(handle, ol))
        self.out_labels = tuple(int(ol[i]) for i in range(nd))
        self.flops = float(lib.ctn_plan_flops(handle))
        self.bytes_min = int(lib.ctn_plan_bytes_min(handle))
        self.out_bytes = int(lib.ctn_plan_out_bytes(handle))

    def workspace_bytes(self, replicas=1):
        return int(self._lib.ctn_plan_workspace_bytes(self._h, replicas))

    def native_step(self, step):
        """Index inside the native plan (its execution order) of the caller's step ``step``."""
        return int(step) if self._native_of is None else int(self._native_of[step])

    def in_caller_order(self, per_native_step, axis=-1):
        """Re-index an array the library filled per native step by the caller's step numbers."""
        if self._native_of is None:
            return per_native_step
        return np.take(per_native_step, self._native_of, axis=axis)

    def step_info(self, step):
        info = StepInfo()
        _check(self._lib.ctn_plan_step_info(self._h, self.native_step(step), C.byref(info)))
        return {name: getattr(info, name) for name, _ in StepInfo._fields_}

    def step_infos(self):
        return [self.step_info(s) for s in range(self.n_steps)]

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.ctn_plan_destroy(h)


class Executor:
    """Device state for ``replicas`` simultaneous contractions of one plan."""

    def __init__(self, plan, replicas=1, device=0, stream=None):
        lib = load_library()
        if device_count() == 0:
            raise RuntimeError(
                "no HIP device visible: the contraction engine has no CPU fallback "
                f"({lib.ctn_last_error().decode()})"
            )
        self.plan = plan
        self.replicas = int(replicas)
        self.device = int(device)
        self._lib = lib
        handle = C.c_void_p()
        _check(lib.ctn_exec_create(plan._h, self.device, C.c_void_p(stream or 0), self.replicas,
                                   C.byref(handle)))
        self._h = handle
        # A native executor owns mutable device state (pointer table, workspace, staging, graph capture) and
        # ctypes releases the GIL inside the native calls: whoever drives it holds this lock from
        # run_host / enqueue through fetch (include/ctn_abi.h, "Threading")
        self.lock = threading.RLock()
        self._log = np.zeros(self.replicas, dtype=np.float64)
        self._resc = np.zeros((self.replicas, plan.n_steps), dtype=np.float64)

    # -- host (numpy) operands ------------------------------------------------
    def run_host(self, operand_sets):
        """``operand_sets``: list (len = replicas) of operand lists (C-contiguous numpy arrays).

        Returns (outs [replicas, *out_shape], device log_scale [replicas], rescales [replicas, n_steps]).
        """
        plan, R = self.plan, self.replicas
        assert len(operand_sets) == R
        with self.lock:
            return self._run_host_locked(operand_sets)

    def _run_host_locked(self, operand_sets):
        plan, R = self.plan, self.replicas
        keep = []
        packed = None
        if R == 1 and plan.n_inputs >= 32 and plan.in_numel_total <= (1 << 20):
            # many small operands (a 1000-matrix chain, a 100-vector hyperedge): ONE concatenate into a buffer of the plan's
            # dtype and the pointers by vector arithmetic, instead of a type check and an address look-up per operand
            # (1001 operands: 1.6 ms of the 3.9 ms call)
            ops = operand_sets[0]
            assert len(ops) == plan.n_inputs
            flat = np.concatenate([np.asarray(o).ravel() for o in ops], dtype=plan.np_dtype, casting="unsafe")
            if flat.size == plan.in_numel_total:
                packed = flat
                addr = (plan.in_offsets_bytes + np.uint64(flat.ctypes.data)).astype(np.uint64)
                keep.append(addr)
                ptrs = (C.c_void_p * plan.n_inputs).from_buffer(addr)
        if packed is None:
            ptrs = (C.c_void_p * (R * plan.n_inputs))()
        for r, ops in enumerate(operand_sets if packed is None else ()):
            assert len(ops) == plan.n_inputs
            for i, op in enumerate(ops):
                if not (isinstance(op, np.ndarray) and op.dtype == plan.np_dtype and op.flags.c_contiguous):
                    op = np.ascontiguousarray(op, dtype=plan.np_dtype)
                    keep.append(op)
                ptrs[r * plan.n_inputs + i] = op.__array_interface__["data"][0]
        outs = np.empty((R,) + plan.out_shape, dtype=plan.np_dtype)
        optrs = (C.c_void_p * R)()
        stride = outs[0].nbytes if R else 0
        for r in range(R):
            optrs[r] = outs.ctypes.data + r * stride
        _check(self._lib.ctn_exec_run(self._h, ptrs, CTN_MEM_HOST, optrs, CTN_MEM_HOST,
                                      _ptr(self._log, C.c_double), _ptr(self._resc, C.c_double)))
        return outs, self._log.copy(), plan.in_caller_order(self._resc).copy()

    # -- device pointers --------------------------------------------------------
    def enqueue(self, in_ptrs, out_ptrs):
        """Asynchronous launch; ``in_ptrs``: replica-major flat list of device addresses."""
        plan, R = self.plan, self.replicas
        assert len(in_ptrs) == R * plan.n_inputs and len(out_ptrs) == R
        ip = (C.c_void_p * len(in_ptrs))(*in_ptrs)
        op = (C.c_void_p * R)(*out_ptrs)
        _check(self._lib.ctn_exec_enqueue(self._h, ip, op))

    def make_enqueue(self, in_ptrs, out_ptrs):
        """Pre-marshal the pointer arrays once; returns a zero-argument launcher."""
        plan, R = self.plan, self.replicas
        assert len(in_ptrs) == R * plan.n_inputs and len(out_ptrs) == R
        ip = (C.c_void_p * len(in_ptrs))(*in_ptrs)
        op = (C.c_void_p * R)(*out_ptrs)
        fn, h = self._lib.ctn_exec_enqueue, self._h

        def launch():
            rc = fn(h, ip, op)
            if rc < 0:
                _check(rc)

        return launch

    def fetch(self):
        _check(self._lib.ctn_exec_fetch(self._h, _ptr(self._log, C.c_double),
                                        _ptr(self._resc, C.c_double)))
        return self._log.copy(), self.plan.in_caller_order(self._resc).copy()

    def synchronize(self):
        _check(self._lib.ctn_exec_synchronize(self._h))

    def set_rescale_mode(self, mode):
        """0 = lazy rescale with overflow detection (default), 1 = eager (the reference's order); returns the
        previous mode."""
        return _check(self._lib.ctn_exec_set_rescale_mode(self._h, int(mode)))

    def set_finish_mode(self, mode):
        """1 = the runs that follow leave the final tensor un-normalised until ``finish`` (include/ctn_abi.h); returns the
        previous mode."""
        return _check(self._lib.ctn_exec_set_finish_mode(self._h, int(mode)))

    def finish(self, mult):
        """Enqueue ``out_r = (out_r / rescale_last_r) * mult[r]`` for a run made in finish mode 1 (after its fetch)."""
        m = np.ascontiguousarray(np.asarray(mult, dtype=np.float64).reshape(self.replicas))
        _check(self._lib.ctn_exec_finish(self._h, _ptr(m, C.c_double)))
        self.synchronize()        # (`m` is read by an asynchronous copy)

    def eager_reruns(self):
        """How often a fetch found a lazily rescaled product out of range and repeated the contraction eagerly."""
        return _check(self._lib.ctn_exec_eager_reruns(self._h))

    # -- device-side join of split-format partial results (include/ctn_abi.h) ------------------
    def snapshot_scales(self, dev_log_ptr, n, host_resc_ptr=0):
        """After an enqueue: copy the log-scale registers of its first ``n`` replicas to device address
        ``dev_log_ptr`` and (optionally) their per-step rescale factors to pinned host address ``host_resc_ptr``,
        asynchronously on the executor's stream."""
        _check(self._lib.ctn_exec_snapshot_scales(self._h, C.c_void_p(dev_log_ptr or None), int(n),
                                                  C.c_void_p(host_resc_ptr or None)))

    def scales_suspect(self, host_resc_ptr, replicas):
        """The range check of ``fetch`` on rescale factors already on the host (``replicas * n_steps`` doubles)."""
        return bool(_check(self._lib.ctn_exec_scales_suspect(self._h, C.c_void_p(host_resc_ptr), int(replicas))))

    def report_suspect(self, suspect):
        """Tell the executor the verdict of a run whose scales the caller checked itself (no fetch): after three suspect
        runs in a row it stays in eager mode.  Returns the streak."""
        return _check(self._lib.ctn_exec_report_suspect(self._h, 1 if suspect else 0))

    def combine_split(self, t_ptr, t_stride, c_ptr, c_stride, n, numel, out_ptr, dtype=None):
        """Enqueue ``out[0:numel], out[numel] = (T_hat, c)`` of ``sum_i t_i exp(c_i)`` over ``n`` parts (device
        pointers; ``t`` of ``dtype``, default the plan's; ``c`` and ``out`` float64)."""
        dt = self.plan.np_dtype if dtype is None else np.dtype(dtype)
        _check(self._lib.ctn_exec_combine_split(self._h, CTN_F32 if dt == np.float32 else CTN_F64, C.c_void_p(t_ptr),
                                                int(t_stride), C.c_void_p(c_ptr), int(c_stride), int(n), int(numel),
                                                C.c_void_p(out_ptr)))

    def add_scales(self, dst_ptr, own_ptr, n, kids=()):
        """Enqueue ``dst[i] = own[i] + sum_j kid_j[idx_j[i]]`` over ``n`` float64 registers (device pointers; ``kids``:
        ``[(scales_ptr, int64_index_ptr)]``, added in order) - the registers of the stages below ride along."""
        k = len(kids)
        sc = (C.c_void_p * max(k, 1))(*[p for p, _ in kids])
        ix = (C.c_void_p * max(k, 1))(*[p for _, p in kids])
        _check(self._lib.ctn_exec_add_scales(self._h, C.c_void_p(dst_ptr), C.c_void_p(own_ptr), int(n), k, sc, ix))

    def merge_scales(self, buf_ptr, stride, numel, scales_ptr, n, extents, merged, dtype=None):
        """Enqueue: the ``n`` evaluations at ``buf + i * stride`` (grid ``extents``, row-major) are brought to a common
        scale along the axes flagged in ``merged`` (include/ctn_abi.h, ctn_exec_merge_scales)."""
        dt = self.plan.np_dtype if dtype is None else np.dtype(dtype)
        ext, mrg = _i32(extents), _i32(1 if m else 0 for m in merged)
        _check(self._lib.ctn_exec_merge_scales(self._h, CTN_F32 if dt == np.float32 else CTN_F64, C.c_void_p(buf_ptr),
                                               int(stride), int(numel), C.c_void_p(scales_ptr), int(n), len(ext),
                                               _ptr(ext, C.c_int32), _ptr(mrg, C.c_int32)))

    def set_timing(self, slots):
        """Bracket every step of the next ``slots`` enqueues with HIP events (0 = off)."""
        _check(self._lib.ctn_exec_set_timing(self._h, int(slots)))

    def step_tiles(self):
        """(tile rows, tile columns) of the MFMA kernel the last enqueue launched per step ((0, 0) otherwise)."""
        out = []
        tm, tn = C.c_int32(0), C.c_int32(0)
        for s in range(self.plan.n_steps):
            _check(self._lib.ctn_exec_step_tile(self._h, self.plan.native_step(s), C.byref(tm), C.byref(tn)))
            out.append((tm.value, tn.value))
        return out

    def step_ms(self):
        ms = np.zeros(self.plan.n_steps, dtype=np.float32)
        _check(self._lib.ctn_exec_step_ms(self._h, _ptr(ms, C.c_float)))
        return self.plan.in_caller_order(ms)

    def is_open(self):
        """False once close() has run (a cached executor evicted by another thread): look it up again."""
        return bool(getattr(self, "_h", None))

    def close(self):
        lock = getattr(self, "lock", None)
        if lock is None:      # __init__ failed before the handle existed
            return
        with lock:            # never under a thread that is still inside run_host / enqueue .. fetch
            h, self._h = getattr(self, "_h", None), None
            if h:
                self._lib.ctn_exec_destroy(h)

    def __del__(self):
        self.close()
