"""``opt_einsum.contract`` names used at reference einsum.py:5, :283, :322, :371-382."""
import numpy as np

_BASE = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ"


def get_symbol(i):
    return _BASE[i] if i < 52 else chr(i + 140)


def parse_backend(arrays, backend):
    if backend != "auto":
        return backend
    for a in arrays:
        return type(a).__module__.split(".")[0]
    return "numpy"


def _tensordot(x, y, axes, backend="numpy"):
    if backend == "torch":      # dispatch only (opt_einsum/backends/torch.py does the same): the arithmetic is torch's
        import torch

        return torch.tensordot(x, y, dims=axes)
    return np.tensordot(x, y, axes=axes)


def _transpose(x, axes, backend="numpy"):
    if backend == "torch":
        return x.permute(*axes)
    return np.transpose(x, axes)


def _einsum(*operands, backend="numpy", **kwargs):
    eq, ops = operands[0], operands[1:]
    table, out = {}, []
    for ch in eq:
        if ch in ",->":
            out.append(ch)
        else:
            table.setdefault(ch, _BASE[len(table)])
            out.append(table[ch])
    if backend == "torch":
        import torch

        return torch.einsum("".join(out), *ops)
    return np.einsum("".join(out), *ops, **kwargs)


def contract_path(subscripts, *shapes, optimize="auto", memory_limit=None, use_blas=True,
                  shapes_flag=None, einsum_call=False, **kw):
    assert kw.pop("shapes", True) and einsum_call
    lhs, out = subscripts.split("->")
    live = lhs.split(",")
    sizes = {}
    for term, shp in zip(live, shapes):
        for s, d in zip(term, shp):
            sizes[s] = d
    n = len(live)
    if isinstance(optimize, str) or optimize is True:
        path = [(0, 1)] + [(0, n - 2 - k) for k in range(n - 2)] if n > 1 else [(0,)]
    else:
        path = [tuple(p) for p in optimize]
    clist = []
    for num, pos in enumerate(path):
        pos = tuple(sorted(pos, reverse=True))
        picked = [live.pop(p) for p in pos]
        needed = set(out)
        for t in live:
            needed |= set(t)
        involved = set("".join(picked))
        removed = involved - needed
        if num == len(path) - 1 and not live:
            result = out
        else:
            result = "".join(s for _, s in sorted((sizes[s], s) for s in involved & needed))
        flag = False
        if len(picked) == 2 and use_blas:
            l, r = picked
            sl, sr = set(l), set(r)
            if (len(sl) == len(l) and len(sr) == len(r) and (sl & sr) == removed
                    and set(result) == (sl | sr) - (sl & sr)):
                flag = "TDOT"
        live.append(result)
        clist.append((pos, removed, ",".join(picked) + "->" + result, None, flag))
    return path, clist
