"""Contraction-path search and the opt_einsum-format contraction list.

The reference delegates this to the third-party ``opt_einsum`` (>=3.3.0,
reference requirements.txt:2; call site reference einsum.py:313-323,
``oe.contract_path(..., shapes=True, einsum_call=True)``).  That package is
absent from this image and from the GPU boxes, so this module restates the
published algorithm of its ``contract_path``:

* explicit paths (a sequence of position tuples) are honoured as given;
* ``"greedy"``: Hadamard-merge operands with identical index sets, then
  repeatedly contract the pair that minimises ``size(out)-size(a)-size(b)``
  (ties broken by smaller flop cost), outer products last;
* ``"optimal"``: exhaustive depth-first search over pairwise orders;
* ``"auto"``/``True``: optimal below 5 operands, greedy otherwise
  (opt_einsum additionally tries branch-and-bound for 5-14 operands).

Every step is emitted as the 5-tuple consumed at reference einsum.py:342:
``(positions_descending, idx_removed, "L,R->O", remaining, blas_flag)`` where
the popped order defines left = higher position, intermediates are ordered by
``(dimension, symbol)`` and the last step uses the caller's output order.

If a real ``opt_einsum`` is importable it is used instead (``use_opt_einsum``).
"""
import itertools

_EINSUM_ONLY = "EINSUM"


def parse_einsum_input(einstr, shapes):
    """Split ``"ab,bc->ac"`` into (input term list, output term, size dict)."""
    if "." in einstr:
        raise ValueError("ellipsis subscripts are not supported")
    if "->" in einstr:
        lhs, out = einstr.split("->")
    else:  # implicit output: symbols appearing once, sorted
        lhs = einstr
        flat = lhs.replace(",", "")
        out = "".join(sorted(s for s in set(flat) if flat.count(s) == 1))
    terms = lhs.split(",")
    if len(terms) != len(shapes):
        raise ValueError(
            f"einsum string has {len(terms)} operands but {len(shapes)} were supplied"
        )
    sizes = {}
    for term, shape in zip(terms, shapes):
        if len(term) != len(shape):
            raise ValueError(
                f"operand with shape {tuple(shape)} does not match subscripts '{term}'"
            )
        for sym, dim in zip(term, shape):
            dim = int(dim)
            old = sizes.get(sym)
            if old is None or old == 1:
                sizes[sym] = dim
            elif dim not in (1, old):
                raise ValueError(f"size of label '{sym}' is inconsistent: {old} vs {dim}")
    for sym in out:
        if sym not in sizes:
            raise ValueError(f"output label '{sym}' does not appear in any operand")
    if len(set(out)) != len(out):
        raise ValueError(f"output subscripts '{out}' repeat a label")
    return terms, out, sizes


def _size(labels, sizes):
    n = 1
    for s in labels:
        n *= sizes[s]
    return n


def _pair_result(a, b, rest_terms, out):
    """Labels kept when contracting index sets a, b given what else still needs them."""
    keep = set(out)
    for t in rest_terms:
        keep |= t
    both = a | b
    return both & keep, both - keep


def _greedy(term_sets, out, sizes):
    """Greedy pairwise path over positions in a shrinking list."""
    live = list(term_sets)
    path = []

    # 1. Hadamard-merge operands carrying identical index sets
    merged = True
    while merged and len(live) > 1:
        merged = False
        seen = {}
        for pos, t in enumerate(live):
            key = frozenset(t)
            if key in seen:
                i, j = seen[key], pos
                rest = [x for k, x in enumerate(live) if k not in (i, j)]
                new, _ = _pair_result(live[i], live[j], rest, out)
                path.append((i, j))
                live = rest + [new]
                merged = True
                break
            seen[key] = pos

    # 2. best connected pair first, outer products only when nothing shares a label.
    #    Label reference counts make "is this label still needed elsewhere" O(1).
    out_set = set(out)
    while len(live) > 1:
        refs = {}
        owners = {}
        for pos, t in enumerate(live):
            for s in t:
                refs[s] = refs.get(s, 0) + 1
                owners.setdefault(s, []).append(pos)
        pairs = set()
        for s, who in owners.items():
            if len(who) > 1:
                pairs.update(itertools.combinations(who, 2))
        if not pairs:  # disconnected pieces: outer product of the two smallest
            order = sorted(range(len(live)), key=lambda p: (_size(live[p], sizes), p))
            pairs = {tuple(sorted(order[:2]))}
        best = None
        for i, j in pairs:
            a, b = live[i], live[j]
            new = {
                s for s in a | b
                if s in out_set or refs[s] - (s in a) - (s in b) > 0
            }
            score = _size(new, sizes) - _size(a, sizes) - _size(b, sizes)
            key = (score, _size(a | b, sizes), i, j)
            if best is None or key < best[0]:
                best = (key, i, j, new)
        _, i, j, new = best
        path.append((i, j))
        live = [x for k, x in enumerate(live) if k not in (i, j)] + [new]
    return path


def _optimal(term_sets, out, sizes):
    """Exhaustive search minimising total flop count (only sensible for few operands)."""
    best = {"cost": None, "path": None}

    def rec(live, path, cost):
        if best["cost"] is not None and cost >= best["cost"]:
            return
        if len(live) == 1:
            best["cost"], best["path"] = cost, list(path)
            return
        for i, j in itertools.combinations(range(len(live)), 2):
            rest = [x for k, x in enumerate(live) if k not in (i, j)]
            new, _ = _pair_result(live[i], live[j], rest, out)
            step = _size(live[i] | live[j], sizes)
            path.append((i, j))
            rec(rest + [new], path, cost + step)
            path.pop()

    rec(list(term_sets), [], 0)
    return best["path"]


def find_path(terms, out, sizes, optimize):
    """Resolve ``optimize`` (strategy name or explicit path) to a list of position tuples."""
    n = len(terms)
    if not isinstance(optimize, (str, bool)) and optimize is not None:
        path = [tuple(int(p) for p in step) for step in optimize]
        if path and path[0] == ("einsum_path",):
            path = path[1:]
        return path
    if n == 1:
        return [(0,)]
    if n == 2:
        return [(0, 1)]
    sets = [set(t) for t in terms]
    name = "auto" if optimize in (True, None) else optimize
    if name is False:
        raise ValueError("optimize=False (single n-ary einsum) is not supported by the HIP engine")
    if name in ("auto", "auto-hq"):
        name = "optimal" if n < 5 else "greedy"
    if name in ("optimal", "dp", "branch-all", "branch-2", "branch-1"):
        if n <= 8:
            return _optimal(sets, out, sizes)
        name = "greedy"
    if name in ("greedy", "eager", "opportunistic") or name.startswith("random-greedy"):
        return _greedy(sets, out, sizes)
    raise KeyError(f"Path optimizer '{optimize}' not found")


def _blas_flag(left, right, result, removed):
    """'TDOT' when the step is a plain tensordot, else an EINSUM-only marker.

    A step is tensordot-able iff no label repeats inside an operand, every
    removed label is shared by both operands and no shared label is kept
    (reference einsum.py:347 routes on this flag; opt_einsum ``can_blas``).
    """
    if len(set(left)) != len(left) or len(set(right)) != len(right):
        return False
    sl, sr = set(left), set(right)
    shared = sl & sr
    if shared != set(removed):
        return False
    if set(result) != (sl | sr) - shared:
        return False
    return "TDOT"


def contraction_list(einstr, shapes, optimize="auto", memory_limit=None, use_blas=True):
    """Restatement of ``oe.contract_path(einstr, *shapes, shapes=True, einsum_call=True)[1]``."""
    terms, out, sizes = parse_einsum_input(einstr, shapes)
    path = find_path(terms, out, sizes, optimize)
    live = list(terms)
    out_set = set(out)
    refs = {}  # label -> number of live terms that carry it
    for t in live:
        for s in set(t):
            refs[s] = refs.get(s, 0) + 1
    steps = []
    for num, positions in enumerate(path):
        positions = tuple(sorted(positions, reverse=True))
        if any(p < 0 or p >= len(live) for p in positions) or len(set(positions)) != len(positions):
            raise ValueError(f"invalid contraction positions {positions} at step {num}")
        picked = [live.pop(p) for p in positions]  # left = highest position
        for t in picked:
            for s in set(t):
                refs[s] -= 1
        involved = set("".join(picked))
        kept = {s for s in involved if s in out_set or refs[s] > 0}
        removed = involved - kept
        last = num == len(path) - 1
        if last and not live:
            result = out
        else:
            result = "".join(s for _, s in sorted((sizes[s], s) for s in kept))
        step_str = ",".join(picked) + "->" + result
        flag = False
        if use_blas and len(picked) == 2:
            flag = _blas_flag(picked[0], picked[1], result, removed)
        live.append(result)
        for s in set(result):
            refs[s] += 1
        steps.append((positions, frozenset(removed), step_str, None, flag))
    if len(live) != 1:
        raise ValueError("contraction path does not reduce the network to a single tensor")
    if steps and sorted(live[0]) != sorted(out):
        raise ValueError("contraction path does not produce the requested output")
    return tuple(steps)


def use_opt_einsum():
    """The real path finder, when the environment has it (it does not here)."""
    try:
        import opt_einsum  # noqa: F401

        return opt_einsum
    except Exception:
        return None


def ssa_to_linear(ssa_path, n_operands):
    """Convert a path written in SSA ids (inputs 0..n-1, step k defines n+k) into
    the shrinking-list positions that ``optimize=`` expects."""
    live = list(range(n_operands))
    linear = []
    for num, step in enumerate(ssa_path):
        pos = tuple(sorted(live.index(t) for t in step))
        for p in reversed(pos):
            live.pop(p)
        live.append(n_operands + num)
        linear.append(pos)
    return tuple(linear)
