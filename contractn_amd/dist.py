"""Multi-GPU execution of the contraction path: one process per GPU, RCCL over xGMI.

The reference has no distributed layer at all.  The path
shards in four ways:

* **replicas** - independent networks (same plan, different tensors) are dealt
  round-robin to the ranks; there is NO data-path collective
  (:func:`shard_range`; this is what ``bench.py --gpus N`` does);
* **batch sharding** - one network with a batch hyperedge (an OUTPUT label every input hangs on: the paper's
  ML workload): every rank takes a chunk of the label's range, no collective on the data path, ONE
  ``all_gather`` of the chunks' results concatenates them (:func:`contract_batch_sharded`);
* **independent subtrees** - one network whose contraction tree is cut near the root: the subtrees are
  dealt to the ranks, their (small) results are exchanged by ONE ``all_gather`` and every rank finishes
  the top of the tree (:func:`contract_subtrees`; for networks with small cuts, e.g. PEPS with D <= 3);
* **index slicing** - one network, a set S of contracted labels is fixed to each
  of its joint values in turn; every slice is an independent contraction with
  the same plan on sliced operands, slices are dealt to the ranks, each rank
  combines its partial results locally in split format and the join is ONE
  ``all_gather`` of ``(T_hat_g, c_g)`` (a scalar PEPS amplitude: 4-8 B + 8 B per
  GPU - latency bound over xGMI, so a single collective and no ring) followed
  by a local log-sum-exp combine.

``torch.distributed`` is plumbing only (backend "nccl" == RCCL on ROCm; "gloo"
in the CPU tests).  The per-slice compute goes through ``contract`` (HIP engine)
unless a different ``contract_fn`` is injected (the CPU tests inject the oracle).
"""
import itertools
import os

import numpy as np

from . import paths


def choose_slices_with_path(einstr, shapes, min_slices=1, max_intermediate=None, max_labels=8, optimize="auto",
                            trials=3):
    """Slice labels and contraction path chosen TOGETHER: repeatedly take the contracted label whose slicing is
    most efficient on the current path - least ``log(work growth) / log(extent)``: 0 is a label every step
    carries, 1 is pure repetition - then search the path again for the sliced sizes (best of ``trials``
    searches), until there are at least ``min_slices`` slices and no intermediate exceeds ``max_intermediate``.
    Returns ``(labels, path, report)``; ``path`` is for the sliced network (what `SlicedContraction` takes as
    ``optimize``).  A fixed path cannot do this (`choose_slices`): on an 8 x 8 PEPS with D = 8 three bonds
    sliced on the row sweep cost 390 x the work, chosen with the path 1.5 x - 512 independent slices whose
    largest intermediate has 2^18 elements instead of 2^27.  Host-only."""
    import math

    shapes = [tuple(int(d) for d in s) for s in shapes]
    terms, out, sizes = paths.parse_einsum_input(einstr, shapes)
    sets = [set(t) for t in terms]

    def search(sz, pool=()):
        # candidates: every path in hand refined for the new sizes (so a step never loses what it had), a fresh
        # search, and refined noisy-greedy trees (a different basin than the fresh search's cluster sweep)
        cands = [paths._reconfigure(sets, out, sz, p, max_leaves=8, rounds=8) for p in pool]
        cands.append(paths.find_path(terms, out, sz, optimize))
        for t in range(trials):
            cands.append(paths._reconfigure(sets, out, sz, paths._random_greedy(sets, out, sz, repeats=4, seed=t),
                                            max_leaves=8, rounds=8))
        ranked, seen = [], set()
        for p in sorted(cands, key=lambda q: paths.path_cost(sets, out, sz, q)):
            key = paths.path_cost(sets, out, sz, p)
            if key not in seen:
                seen.add(key)
                ranked.append(p)
        return paths.path_cost(sets, out, sz, ranked[0]), ranked[:3]

    (base_flops, base_big), pool = search(sizes)
    path = pool[0]
    sz, count, chosen = dict(sizes), 1, []
    flops, big = base_flops, base_big
    while (count < min_slices or (max_intermediate is not None and big > max_intermediate)) and len(chosen) < max_labels:
        best = None
        for lab in sorted(sizes):
            if lab in out or sz[lab] == 1 or sizes[lab] < 2:
                continue
            s2 = dict(sz)
            s2[lab] = 1
            # the label's efficiency on the best of the paths in hand
            f, b = min(paths.path_cost(sets, out, s2, p) for p in pool)
            eff = math.log(max(f * sizes[lab], 1) / max(flops, 1)) / math.log(sizes[lab])
            key = (eff, b, lab)
            if best is None or key < best[0]:
                best = (key, lab)
        if best is None:
            break
        lab = best[1]
        chosen.append(lab)
        sz[lab] = 1
        count *= sizes[lab]
        (flops, big), pool = search(sz, pool)
        path = pool[0]
    # the path indexes the operands of the sliced network, which are the same operands in the same order
    report = {"slices": count, "largest_intermediate": big, "unsliced_largest_intermediate": base_big,
              "work_overhead": flops * count / max(base_flops, 1), "unsliced_flops": base_flops}
    return tuple(chosen), tuple(tuple(p) for p in path), report


def sliced_plan(einstr, shapes, min_slices=1, max_intermediate=None, cache_dir=None, **kwargs):
    """`choose_slices_with_path` behind a file cache: the search takes tens of seconds on the host and depends
    only on the network's structure, so its result ``(labels, path, report)`` is kept as JSON under
    ``cache_dir`` (default ``contractn_amd/plans/``), keyed by a hash of (einsum string, shapes, targets).
    A cached entry is checked against the network before use (labels contracted, path reduces the operands)."""
    import hashlib
    import json
    import os

    shapes = [tuple(int(d) for d in s) for s in shapes]
    cache_dir = cache_dir or os.path.join(os.path.dirname(os.path.abspath(__file__)), "plans")
    key = hashlib.sha1(json.dumps([einstr, shapes, int(min_slices), max_intermediate, sorted(kwargs.items())],
                                  ensure_ascii=True).encode()).hexdigest()[:16]
    fname = os.path.join(cache_dir, f"sliced_{key}.json")
    if os.path.exists(fname):
        try:
            with open(fname) as fh:
                d = json.load(fh)
            labels = tuple(d["labels"])
            path = tuple(tuple(int(x) for x in p) for p in d["path"])
            lhs, out = einstr.split("->")
            if (d.get("einsum_str") == einstr and all(lab in lhs and lab not in out for lab in labels)
                    and len(path) == len(shapes) - 1):
                return labels, path, d["report"]
        except (OSError, ValueError, KeyError):
            pass
    labels, path, report = choose_slices_with_path(einstr, shapes, min_slices=min_slices,
                                                   max_intermediate=max_intermediate, **kwargs)
    try:
        os.makedirs(cache_dir, exist_ok=True)
        with open(fname, "w") as fh:
            json.dump({"einsum_str": einstr, "shapes": shapes, "min_slices": int(min_slices),
                       "max_intermediate": max_intermediate, "labels": list(labels),
                       "path": [list(p) for p in path], "report": report}, fh)
    except OSError:
        pass   # read-only installation: search again next time
    return labels, path, report


def shard_range(n_items, rank, world):
    """Contiguous, balanced share of ``range(n_items)`` for ``rank``."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def combine_split(parts):
    """Sum tensors given in split format ``[(T_hat, c), ...]`` without leaving it.

    ``sum_g T_hat_g * exp(c_g) = exp(c*) * sum_g T_hat_g * exp(c_g - c*)`` with
    ``c* = max c_g``; the result is re-stabilised (mean |T| == 1) exactly like a
    contraction step (reference einsum.py:89-107).
    """
    parts = [(np.asarray(t), float(c)) for t, c in parts]
    live = [(t, c) for t, c in parts if np.any(t != 0)]
    if not live:
        return parts[0][0] * 0, np.zeros(())
    c_star = max(c for _, c in live)
    total = sum(t.astype(np.float64) * np.exp(c - c_star) for t, c in live)
    norm = np.sum(np.abs(total))
    if norm > 1e-7:
        rescale = norm / total.size
        total = total / rescale
        c_star = c_star + np.log(rescale)
    return total.astype(parts[0][0].dtype), np.asarray(c_star, dtype=np.float64)


def choose_slices(einstr, shapes, optimize="auto", max_intermediate=None, min_slices=1, max_labels=12):
    """Pick the labels to slice for a given path: greedily the contracted label whose removal shrinks the
    largest intermediate most (ties: the least element count left above the target, then the least
    total work; once the memory target is met, simply the least added work), until the largest
    intermediate is at most ``max_intermediate`` elements AND there are at least ``min_slices`` slices
    (e.g. one per GPU).  Returns ``(labels, report)`` with the largest intermediate, the slice count and the
    work overhead ``total sliced flop proxy / unsliced flop proxy`` - slicing a label that only part of the
    path carries repeats the rest of the path for every value, so the overhead tells when a different
    path (``optimize``) should be tried before more labels are sliced.  Host-only; the path is fixed."""
    shapes = [tuple(int(d) for d in s) for s in shapes]
    terms, out, sizes = paths.parse_einsum_input(einstr, shapes)
    path = paths.find_path(terms, out, sizes, optimize)
    sets = [set(t) for t in terms]
    base_flops, base_big = paths.path_cost(sets, out, sizes, path)

    def evaluate(labels):
        sz = dict(sizes)
        count = 1
        for lab in labels:
            count *= sizes[lab]
            sz[lab] = 1
        flops, inter = paths.path_profile(sets, out, sz, path)
        big = max(inter, default=1)
        # what still sticks out above the target (all of it when there is no target): the greedy step
        # needs a measure that moves even when no single label lowers the peak
        excess = sum(x for x in inter if max_intermediate is None or x > max_intermediate)
        return flops * count, big, count, excess

    chosen = []
    flops, big, count = base_flops, base_big, 1
    candidates = sorted(s for s in sizes if s not in out and sizes[s] > 1)
    while (max_intermediate is not None and big > max_intermediate) or count < min_slices:
        if len(chosen) >= max_labels:
            break
        best = None
        for lab in candidates:
            if lab in chosen:
                continue
            f, b, c, x = evaluate(chosen + [lab])
            key = (b, x, f, lab) if max_intermediate is not None and big > max_intermediate else (f, lab)
            if best is None or key < best[0]:
                best = (key, lab, f, b, c)
        if best is None:
            break
        _, lab, flops, big, count = best
        chosen.append(lab)
    return tuple(chosen), {"slices": count, "largest_intermediate": big, "unsliced_largest_intermediate": base_big,
                           "work_overhead": flops / max(base_flops, 1)}


def slice_network(einstr, operands, slice_labels):
    """Yield ``(index tuple, sliced einsum string, sliced operands)`` for every joint
    value of ``slice_labels`` (labels must be contracted, i.e. absent from the output)."""
    shapes = [np.shape(o) for o in operands]
    terms, out, sizes = paths.parse_einsum_input(einstr, shapes)
    for lab in slice_labels:
        if lab in out:
            raise ValueError(f"cannot slice over output label '{lab}'")
        if lab not in sizes:
            raise ValueError(f"label '{lab}' is not in the network")
    new_terms = ["".join(s for s in t if s not in slice_labels) for t in terms]
    sliced_str = ",".join(new_terms) + "->" + out
    ranges = [range(sizes[lab]) for lab in slice_labels]
    for values in itertools.product(*ranges):
        fix = dict(zip(slice_labels, values))
        ops = []
        for t, op in zip(terms, operands):
            idx = tuple(fix[s] if s in fix else slice(None) for s in t)
            ops.append(np.ascontiguousarray(np.asarray(op)[idx]))
        yield values, sliced_str, ops


def n_slices(einstr, operands, slice_labels):
    shapes = [np.shape(o) for o in operands]
    _, _, sizes = paths.parse_einsum_input(einstr, shapes)
    n = 1
    for lab in slice_labels:
        n *= sizes[lab]
    return n


def contract_sliced(einstr, operands, slice_labels, optimize="auto", contract_fn=None,
                    group=None, rank=None, world=None):
    """Sliced contraction across the ranks of ``group``; returns ``(T_hat, log_scale)`` on every rank.

    With ``world == 1`` (or torch.distributed not initialised) all slices run
    locally and no collective is issued.
    """
    if contract_fn is None:
        from .einsum import contract as contract_fn
    dist = None
    if world is None or rank is None:
        try:
            import torch.distributed as dist_mod

            if dist_mod.is_available() and dist_mod.is_initialized():
                dist = dist_mod
                world = dist.get_world_size(group)
                rank = dist.get_rank(group)
        except ImportError:
            pass
    if world is None:
        world, rank = 1, 0

    total = n_slices(einstr, operands, slice_labels)
    mine = set(shard_range(total, rank, world))
    local = []
    sliced_path = None
    for num, (_vals, sliced_str, ops) in enumerate(slice_network(einstr, operands, slice_labels)):
        if num not in mine:
            continue
        if sliced_path is None:
            sliced_path = optimize
        local.append(contract_fn(sliced_str, *ops, optimize=sliced_path, split_format=True))
    out_shape = None
    if local:
        t_loc, c_loc = combine_split(local)
        out_shape = t_loc.shape
    else:  # more ranks than slices: contribute an exact zero
        shapes = [np.shape(o) for o in operands]
        _, out, sizes = paths.parse_einsum_input(einstr, shapes)
        out_shape = tuple(sizes[s] for s in out)
        t_loc, c_loc = np.zeros(out_shape, dtype=np.result_type(*[np.asarray(o).dtype for o in operands])), np.zeros(())
    if world == 1:
        return t_loc, c_loc
    return all_gather_combine(t_loc, c_loc, group=group, world=world)


def all_gather_combine(t_loc, c_loc, group=None, world=None, device=None):
    """THE join: one all_gather of the packed ``(T_hat, c)`` buffers, then a local combine.

    ``device``: the GPU this rank computes on (RCCL needs the buffers there); defaults to torch's current
    device.  With the gloo backend (CPU tests, one-GPU rehearsals) the buffers stay on the host."""
    import torch
    import torch.distributed as dist

    world = world or dist.get_world_size(group)
    backend = dist.get_backend(group)
    if backend == "nccl":
        device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
    else:
        device = torch.device("cpu")
    flat = np.concatenate([np.asarray(t_loc, dtype=np.float64).ravel(), [float(c_loc)]])
    send = torch.from_numpy(flat).to(device)
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send, group=group)
    parts = []
    for buf in recv:
        arr = buf.cpu().numpy()
        parts.append((arr[:-1].reshape(np.shape(t_loc)).astype(np.asarray(t_loc).dtype), arr[-1]))
    return combine_split(parts)


# ---------------------------------------------------------------------------
# batch sharding ("batched workloads ... are embarrassingly data-parallel: replicas only,
# concatenate outputs")
# ---------------------------------------------------------------------------
def concat_split(parts, axis):
    """Concatenate tensors given in split format ``[(T_hat, c), ...]`` along ``axis`` without leaving it:
    ``T = concat_g T_hat_g exp(c_g - c*)`` with ``c* = max c_g`` over the non-zero pieces, re-stabilised like a
    contraction step (mean |T_hat| == 1; reference einsum.py:89-107)."""
    parts = [(np.asarray(t), float(c)) for t, c in parts]
    live = [c for t, c in parts if np.any(t != 0)]
    dtype = parts[0][0].dtype
    if not live:
        return np.concatenate([t for t, _ in parts], axis=axis), np.zeros(())
    c_star = max(live)
    total = np.concatenate([t.astype(np.float64) * (np.exp(c - c_star) if np.any(t != 0) else 0.0) for t, c in parts], axis=axis)
    norm = np.sum(np.abs(total))
    if norm > 1e-7:
        rescale = norm / total.size
        total = total / rescale
        c_star = c_star + np.log(rescale)
    return total.astype(dtype), np.asarray(c_star, dtype=np.float64)


def shard_batch_label(einstr, operands, label, rank, world):
    """This rank's share of a data-parallel contraction over the OUTPUT label ``label`` (a batch hyperedge: the
    copy node every input hangs on): its contiguous chunk ``[lo, hi)`` of the label's range, with every operand that
    carries the label cut down to it (views, no copies) and the others passed through.  Returns
    ``(operands, lo, hi, out_axis)``; the einsum string is unchanged."""
    lhs, out = einstr.split("->")
    terms = lhs.split(",")
    if label not in out:
        raise ValueError(f"'{label}' is summed: shard it with SlicedContraction, not as a batch")
    shapes = [tuple(o.shape) for o in operands]
    _, _, sizes = paths.parse_einsum_input(einstr, shapes)
    mine_r = shard_range(sizes[label], rank, world)
    lo, hi = mine_r.start, mine_r.stop
    if hi <= lo:
        raise ValueError(f"rank {rank} of {world} gets no part of label '{label}' (extent {sizes[label]})")
    mine = []
    for t, op in zip(terms, operands):
        idx = tuple(slice(lo, hi) if s == label else slice(None) for s in t)
        mine.append(op[idx] if label in t else op)
    return mine, lo, hi, out.index(label)


def contract_batch_sharded(einstr, operands, label, optimize="auto", contract_fn=None, group=None, rank=None,
                           world=None, device=None):
    """Data-parallel contraction over a batch label: rank g contracts chunk g of the label's range on its own
    GPU - no collective on the data path - and ONE all_gather of the chunks' split-format results ``(T_hat_g, c_g)``
    joins them into the full ``(T_hat, c)`` on every rank (`concat_split`).  ``contract_fn(einstr, *ops,
    optimize=..., split_format=True)`` defaults to the HIP engine.  NumPy operands and results (a classifier's
    outputs are B numbers); the chunks' operands may be any strided views."""
    import torch
    import torch.distributed as dist

    if contract_fn is None:
        from .einsum import contract as contract_fn
    distributed = dist.is_available() and dist.is_initialized()
    rank = (dist.get_rank(group) if distributed else 0) if rank is None else rank
    world = (dist.get_world_size(group) if distributed else 1) if world is None else world
    # (checked on EVERY rank before anything is contracted: a rank with an empty chunk must not raise alone while
    # the others wait for it in the all_gather)
    _sizes = paths.parse_einsum_input(einstr, [tuple(o.shape) for o in operands])[2]
    if label in _sizes and _sizes[label] < world:
        raise ValueError(f"batch label '{label}' has extent {_sizes[label]} < {world} ranks: some rank would get no part")
    mine, lo, hi, axis = shard_batch_label(einstr, operands, label, rank, world)
    mine = [np.ascontiguousarray(o) for o in mine]
    t_loc, c_loc = contract_fn(einstr, *mine, optimize=optimize, split_format=True)
    t_loc = np.asarray(t_loc)
    if world == 1 or not distributed:
        return t_loc, np.asarray(float(c_loc), dtype=np.float64)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device() if device is None else device) if backend == "nccl" else torch.device("cpu")
    shapes = [tuple(o.shape) for o in operands]
    _, _, sizes = paths.parse_einsum_input(einstr, shapes)
    # chunks may differ by one index: pad every rank's buffer to the largest chunk
    per = t_loc.size // (hi - lo)
    widest = max(len(shard_range(sizes[label], g, world)) for g in range(world))
    flat = np.zeros(widest * per + 1, dtype=np.float64)
    flat[:t_loc.size] = np.moveaxis(t_loc, axis, 0).astype(np.float64).ravel()
    flat[-1] = float(c_loc)
    send = torch.from_numpy(flat).to(dev)
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send, group=group)
    parts = []
    rest = tuple(np.delete(np.array(t_loc.shape), axis))
    for g, buf in enumerate(recv):
        arr = buf.cpu().numpy()
        n_g = len(shard_range(sizes[label], g, world))
        piece = arr[:n_g * per].reshape((n_g,) + tuple(int(x) for x in rest))
        parts.append((np.moveaxis(piece, 0, axis).astype(t_loc.dtype), arr[-1]))
    return concat_split(parts, axis)


# ---------------------------------------------------------------------------
# independent subtrees (north_star: "independent subtrees of the contraction path shard across the GPUs of one
# node with a single RCCL reduce/all-gather over xGMI at the join")
# ---------------------------------------------------------------------------
def subtree_plan(einstr, shapes, optimize="auto", n_parts=2, max_boundary=1 << 22):
    """Cut the contraction tree of ``optimize`` into independent subtrees plus the top part that joins them.

    The tree is opened from the root: the most expensive open node is replaced by its two children until there
    are at least ``2 * n_parts`` open nodes (finer pieces balance better) or no node can be opened without an
    open node's result exceeding ``max_boundary`` elements - the results are what crosses the links at the join,
    so this mode is for networks whose cuts are small (a PEPS half has D^L boundary elements; fine
    for D <= 3, hopeless for D >= 8 - those shard by index slicing, `SlicedContraction`).

    Returns ``(parts, top)``:
      parts[i] = dict(operands=[indices into the network's operands], einsum=..., path=linear path, cost=...,
                      out=labels of the subtree's result)   - a single-operand part has ``path == ()``
      top      = dict(einsum=..., path=linear path over the parts' results in order)
    """
    shapes = [tuple(int(d) for d in s) for s in shapes]
    terms, out, sizes = paths.parse_einsum_input(einstr, shapes)
    n = len(terms)
    sets = [set(t) for t in terms]
    if isinstance(optimize, str) or optimize is True:
        # no tree given: a searched path is usually a sweep (a caterpillar tree has no balanced cut), so the
        # operands are first split into balanced, weakly connected groups and each group gets its own path
        groups = _partition_operands(sets, sizes, out, n_parts, max_boundary)
        if len(groups) > 1:
            how = "auto" if optimize is True else optimize
            parts = []
            for g in groups:
                inside = set(g)
                mine = set().union(*[sets[i] for i in g])
                outside = set(out).union(*[sets[i] for i in range(n) if i not in inside])
                lab = mine & outside
                out_term = "".join([c for c in out if c in lab] + sorted(lab - set(out)))
                sub_terms = [terms[i] for i in g]
                sub_path = tuple(tuple(p) for p in paths.find_path(sub_terms, out_term, sizes, how)) if len(g) > 1 else ()
                cst = paths.path_cost([set(t) for t in sub_terms], out_term, sizes, sub_path)[0] if len(g) > 1 else 0
                parts.append({"operands": list(g), "einsum": ",".join(sub_terms) + "->" + out_term, "path": sub_path,
                              "cost": cst, "out": out_term})
            top_terms = [p["out"] for p in parts]
            top = {"einsum": ",".join(top_terms) + "->" + out,
                   "path": tuple(tuple(p) for p in paths.find_path(top_terms, out, sizes, how))}
            return parts, top
        lin = paths.find_path(terms, out, sizes, optimize)
    else:
        lin = optimize
    # linear positions -> SSA tree
    live = list(range(n))
    children, leaves, cost = {}, {i: (i,) for i in range(n)}, {i: 0 for i in range(n)}
    for num, step in enumerate(lin):
        ids = [live[p] for p in sorted(step)]
        for p in sorted(step, reverse=True):
            live.pop(p)
        new = n + num
        if len(ids) == 1:        # unary step (trace / sum-out): stays with its operand's subtree
            children[new] = (ids[0],)
            leaves[new] = leaves[ids[0]]
            cost[new] = cost[ids[0]]
        else:
            children[new] = tuple(ids)
            leaves[new] = tuple(sorted(leaves[ids[0]] + leaves[ids[1]]))
            joint = set().union(*[sets[i] for i in leaves[new]])
            cost[new] = cost[ids[0]] + cost[ids[1]] + paths._size(
                labels_of(ids[0], leaves, sets, out, n) | labels_of(ids[1], leaves, sets, out, n), sizes)
            del joint
        live.append(new)
    root = live[0]

    def out_size(node):
        return paths._size(labels_of(node, leaves, sets, out, n), sizes)

    open_nodes = [root]
    while len(open_nodes) < 2 * n_parts:
        cands = [x for x in open_nodes if x >= n and len(children[x]) == 2
                 and all(out_size(c) <= max_boundary for c in children[x])]
        if not cands:
            break
        pick = max(cands, key=lambda x: cost[x])
        at = open_nodes.index(pick)
        open_nodes[at:at + 1] = list(children[pick])
    open_set = set(open_nodes)

    def local_path(node_root, operand_ids):
        """SSA steps below ``node_root`` (stopping at ``operand_ids``), renumbered over that operand list."""
        local = {x: i for i, x in enumerate(operand_ids)}
        steps = []

        def walk(x):
            if x in local:
                return local[x]
            kids = [walk(c) for c in children[x]]
            local[x] = len(operand_ids) + len(steps)
            steps.append(tuple(kids))
            return local[x]

        walk(node_root)
        return paths.ssa_to_linear(steps, len(operand_ids)) if steps else ()

    parts = []
    for x in open_nodes:
        ops = list(leaves[x])
        lab = labels_of(x, leaves, sets, out, n)
        # keep the caller's output order for labels that survive to the end, then the rest sorted
        out_term = "".join([c for c in out if c in lab] + sorted(lab - set(out)))
        parts.append({"operands": ops, "einsum": ",".join(terms[i] for i in ops) + "->" + out_term,
                      "path": local_path(x, ops), "cost": cost[x], "out": out_term})
    top = {"einsum": ",".join(p["out"] for p in parts) + "->" + out,
           "path": local_path(root, open_nodes) if root not in open_set else ()}
    return parts, top


def _partition_operands(sets, sizes, out, n_parts, max_boundary):
    """Balanced groups of operands with small cuts: recursive bisection of the network graph (operands joined by
    shared labels) - a region grown breadth-first from a peripheral operand until it holds half the weight, then
    boundary operands moved across while that shrinks the cut (weight = log2 of an operand's size; a group whose
    open legs would exceed ``max_boundary`` elements is not split further).  Deterministic."""
    import math

    n = len(sets)
    weight = [1.0 + math.log2(max(paths._size(s, sizes), 1)) for s in sets]
    by_label = {}
    for i, s in enumerate(sets):
        for c in s:
            by_label.setdefault(c, []).append(i)
    nbrs = [sorted({j for c in sets[i] for j in by_label[c] if j != i}) for i in range(n)]

    def boundary(group):
        inside = set(group)
        mine = set().union(*[sets[i] for i in group])
        outside = set(out).union(*[sets[i] for i in range(n) if i not in inside])
        return paths._size(mine & outside, sizes)

    def bfs_order(group, start):
        inside, seen, order, queue = set(group), {start}, [], [start]
        while queue:
            x = queue.pop(0)
            order.append(x)
            for y in nbrs[x]:
                if y in inside and y not in seen:
                    seen.add(y)
                    queue.append(y)
        return order + [x for x in group if x not in seen]     # disconnected leftovers last

    def bisect(group):
        far = bfs_order(group, group[0])[-1]                    # a peripheral operand
        order = bfs_order(group, far)
        total, acc, left = sum(weight[i] for i in group), 0.0, []
        for x in order:
            if acc >= total / 2 and left:
                break
            left.append(x)
            acc += weight[x]
        a, b = set(left), set(group) - set(left)
        if not b:
            return None
        for _ in range(4):                                      # a few refinement sweeps
            moved = False
            for x in sorted(group):
                src, dst = (a, b) if x in a else (b, a)
                if len(src) <= 1:
                    continue
                wa = sum(weight[i] for i in a)
                new_wa = wa - weight[x] if x in a else wa + weight[x]
                if abs(new_wa - total / 2) > 0.15 * total + max(weight):
                    continue
                before = boundary(sorted(a)) * boundary(sorted(b))
                src.discard(x)
                dst.add(x)
                if boundary(sorted(a)) * boundary(sorted(b)) < before:
                    moved = True
                else:
                    dst.discard(x)
                    src.add(x)
            if not moved:
                break
        return sorted(a), sorted(b)

    groups = [list(range(n))]
    while len(groups) < n_parts:
        cands = sorted((g for g in groups if len(g) > 1), key=lambda g: -sum(weight[i] for i in g))
        done = False
        for g in cands:
            halves = bisect(g)
            if halves and all(boundary(h) <= max_boundary for h in halves):
                groups.remove(g)
                groups.extend(halves)
                done = True
                break
        if not done:
            break
    return sorted(groups)


def labels_of(node, leaves, sets, out, n):
    """Labels of a subtree's result: those of its leaves that are also needed outside it (or in the output)."""
    inside = set(leaves[node])
    mine = set().union(*[sets[i] for i in inside])
    outside = set(out)
    for i in range(n):
        if i not in inside:
            outside |= sets[i]
    return mine & outside


def assign_parts(costs, world):
    """Longest-processing-time assignment of subtree costs to ranks; returns the rank of every part."""
    load = [0] * world
    owner = [0] * len(costs)
    for i in sorted(range(len(costs)), key=lambda i: (-costs[i], i)):
        r = min(range(world), key=lambda r: (load[r], r))
        owner[i] = r
        load[r] += costs[i]
    return owner


def contract_subtrees(einstr, operands, optimize="auto", contract_fn=None, group=None, rank=None, world=None,
                      max_boundary=1 << 22, device=None, n_parts=None):
    """Contract one network with its independent subtrees dealt to the ranks of ``group``; returns
    ``(T_hat, log_scale)`` on every rank.

    Every rank contracts the subtrees it owns (HIP engine unless ``contract_fn`` is injected), the results - in
    split format, so nothing can overflow on the way - are exchanged by ONE ``all_gather`` of a packed buffer, and
    every rank then runs the small top part of the tree on the gathered results: ``c = c_top + sum c_subtree``.
    """
    if contract_fn is None:
        from .einsum import contract as contract_fn
    dist = None
    if world is None or rank is None:
        try:
            import torch.distributed as dist_mod

            if dist_mod.is_available() and dist_mod.is_initialized():
                dist = dist_mod
                world, rank = dist.get_world_size(group), dist.get_rank(group)
        except ImportError:
            pass
    if world is None:
        world, rank = 1, 0
    shapes = [np.shape(o) for o in operands]
    parts, top = subtree_plan(einstr, shapes, optimize=optimize, n_parts=n_parts or world, max_boundary=max_boundary)
    owner = assign_parts([p["cost"] for p in parts], world)
    dtype = np.result_type(*[np.asarray(o).dtype for o in operands])
    _, _, sizes = paths.parse_einsum_input(einstr, shapes)
    numels = [int(np.prod([sizes[c] for c in p["out"]])) if p["out"] else 1 for p in parts]

    results = [None] * len(parts)
    for i, p in enumerate(parts):
        if owner[i] != rank:
            continue
        ops = [operands[j] for j in p["operands"]]
        if len(ops) == 1 and p["einsum"].split("->")[0] == p["out"]:
            results[i] = (np.asarray(ops[0], dtype=dtype), 0.0)     # a bare operand: nothing to contract
        else:
            t, c = contract_fn(p["einsum"], *ops, optimize=p["path"] if len(ops) > 1 else "auto", split_format=True)
            results[i] = (np.asarray(t), float(c))
    if world > 1:
        import torch

        if dist is None:
            import torch.distributed as dist
        # one packed segment per rank: [T_hat of its parts ..., their log-scales ...], padded to the longest
        seg = [sum(numels[i] + 1 for i in range(len(parts)) if owner[i] == r) for r in range(world)]
        buf = np.zeros(max(seg), dtype=np.float64)
        pos = 0
        for i in range(len(parts)):
            if owner[i] == rank:
                buf[pos:pos + numels[i]] = np.asarray(results[i][0], dtype=np.float64).ravel()
                buf[pos + numels[i]] = results[i][1]
                pos += numels[i] + 1
        backend = dist.get_backend(group)
        dev = (torch.device("cuda", torch.cuda.current_device() if device is None else device)
               if backend == "nccl" else torch.device("cpu"))
        send = torch.from_numpy(buf).to(dev)
        recv = [torch.empty_like(send) for _ in range(world)]
        dist.all_gather(recv, send, group=group)              # THE join
        for r in range(world):
            arr = recv[r].cpu().numpy()
            pos = 0
            for i in range(len(parts)):
                if owner[i] == r:
                    shp = tuple(sizes[c] for c in parts[i]["out"])
                    results[i] = (arr[pos:pos + numels[i]].reshape(shp).astype(dtype), float(arr[pos + numels[i]]))
                    pos += numels[i] + 1
    c_sub = sum(c for _, c in results)
    if len(parts) == 1:
        return results[0][0], np.asarray(c_sub)
    t, c = contract_fn(top["einsum"], *[t for t, _ in results], optimize=top["path"], split_format=True)
    return t, np.asarray(float(c) + c_sub)


def _hashable(optimize):
    if isinstance(optimize, (str, bool)) or optimize is None:
        return "auto" if optimize is True else optimize
    return tuple(tuple(int(p) for p in step) for step in optimize)


class SlicedContraction:
    """Device-resident index slicing: every slice is one *replica* of the sliced plan.

    The operands are uploaded once; a slice fixes the sliced labels, which for a tensor whose
    sliced axes lead is just a pointer offset - so the slices of this rank run as ``R`` replicas of
    ONE plan in one launch sequence (zero-copy), each producing ``(T_hat_s, c_s)``.  The partial
    results are combined in split format on the host and joined across ranks with the single
    ``all_gather`` of :func:`all_gather_combine`.
    """

    def __init__(self, einstr, operands, slice_labels, optimize="auto", rank=0, world=1, device=0,
                 dtype=None, workspace_budget=64 << 30, slices=None):
        """``slices``: the joint values of ``slice_labels`` this rank contracts (tuples), instead of its contiguous
        share of all of them - for a caller that has dealt the slices itself (`StagedSlicedContraction` cuts the grid of
        label values into blocks; its checked fallback must cover exactly its own block)."""
        import torch

        from . import einsum as E

        self.rank, self.world = rank, world
        shapes = [tuple(np.shape(o)) for o in operands]
        terms, out, sizes = paths.parse_einsum_input(einstr, shapes)
        for lab in slice_labels:
            if lab in out or lab not in sizes:
                raise ValueError(f"cannot slice over label '{lab}'")
        self.slice_labels = tuple(slice_labels)
        self.np_dtype = np.dtype(dtype or np.result_type(*[np.asarray(o).dtype for o in operands]))
        if self.np_dtype not in (np.float32, np.float64):
            self.np_dtype = np.dtype(np.float64)
        tdt = torch.float32 if self.np_dtype == np.float32 else torch.float64
        dev = torch.device("cuda", device)
        # sliced axes first, so a slice of an operand is one contiguous block at a pointer offset
        self.tensors, self.perm_terms, new_terms, new_shapes = [], [], [], []
        for term, op in zip(terms, operands):
            lead = [i for i, s in enumerate(term) if s in self.slice_labels]
            rest = [i for i, s in enumerate(term) if s not in self.slice_labels]
            t = torch.as_tensor(np.ascontiguousarray(np.transpose(np.asarray(op, dtype=self.np_dtype), lead + rest)),
                                device=dev, dtype=tdt)
            self.tensors.append(t)
            self.perm_terms.append("".join(term[i] for i in lead))
            new_terms.append("".join(term[i] for i in rest))
            new_shapes.append(tuple(sizes[term[i]] for i in rest))
        self.sliced_str = ",".join(new_terms) + "->" + out
        self.sizes = sizes
        ranges = [range(sizes[lab]) for lab in self.slice_labels]
        all_slices = list(itertools.product(*ranges))
        if slices is None:
            self.my_slices = [all_slices[i] for i in shard_range(len(all_slices), rank, world)]
        else:
            self.my_slices = [tuple(int(x) for x in v) for v in slices]
            if any(len(v) != len(ranges) or any(x not in rg for x, rg in zip(v, ranges)) for v in self.my_slices):
                raise ValueError("slices must be joint values of the sliced labels")
        self.out_shape = tuple(sizes[s] for s in out)
        self.n_total = len(all_slices)
        self.device = device
        self.bc = None
        self._chunks = []
        if self.my_slices:
            # slices run in groups of R replicas; R is bounded by the workspace the plan needs
            probe = E._native_plan(E._contract_path(self.sliced_str, tuple(new_shapes), optimize=_hashable(optimize),
                                                    memory_limit=None, use_blas=True),
                                   tuple(new_shapes), self.np_dtype.name)
            per = max(1, probe.workspace_bytes(2) - probe.workspace_bytes(1))
            R = int(max(1, min(len(self.my_slices), workspace_budget // per)))
            self.bc = E.BatchedContraction(self.sliced_str, new_shapes, self.np_dtype, optimize=optimize,
                                           replicas=R, device=device)
            item = self.np_dtype.itemsize
            self._owned = []

            def slice_ptrs(values):
                fix = dict(zip(self.slice_labels, values))
                ptrs = []
                for t, lead_term, shp in zip(self.tensors, self.perm_terms, new_shapes):
                    block = int(np.prod(shp)) if shp else 1
                    idx = 0
                    for s in lead_term:
                        idx = idx * sizes[s] + fix[s]
                    p = t.data_ptr() + idx * block * item
                    if p % 16:  # odd-sized block: materialise an aligned copy of this slice
                        sl = t[tuple(fix[s] for s in lead_term)].clone()
                        self._owned.append(sl)
                        p = sl.data_ptr()
                    ptrs.append(p)
                return ptrs

            self.out = torch.zeros((len(self.my_slices),) + self.out_shape, device=dev, dtype=tdt)
            self._scratch_out = torch.zeros((R,) + self.out_shape, device=dev, dtype=tdt)
            for c0 in range(0, len(self.my_slices), R):
                group = self.my_slices[c0:c0 + R]
                n = len(group)
                padded = group + [group[-1]] * (R - n)   # short last group: repeat a slice, ignore its output
                in_ptrs = [p for values in padded for p in slice_ptrs(values)]
                out_ptrs = [self.out[c0 + r].data_ptr() if r < n else self._scratch_out[r].data_ptr()
                            for r in range(R)]
                self._chunks.append((c0, n, self.bc.executor.make_enqueue(in_ptrs, out_ptrs)))
            self.R = R
            # everything above was prepared on torch's current stream (uploads, clones of odd-sized slices, the
            # zero fills); the executor runs on its own non-blocking stream, which does not order against it
            torch.cuda.current_stream(dev).synchronize()
        self.last_slices = None   # (T_hat [n_local, ...], log_scale [n_local]) of the last local_result()
        self._join = None         # device / pinned buffers of run_small, allocated on first use
        self.root_members = [[i] for i in range(len(self.my_slices))]   # (uniform with StagedSlicedContraction)

    def slices_host(self):
        """``(T_hat [n_local, ...], c [n_local])`` of this rank's slices, through the checked host path."""
        self.local_result()
        return self.last_slices

    def stage_list(self):
        """``[(BatchedContraction, evaluations on this rank, replicas per launch, launches per contraction)]`` - one
        entry here; `StagedSlicedContraction` has one per stage (measurement tooling: bench.py)."""
        return [] if self.bc is None else [(self.bc, len(self.my_slices), self.R, len(self._chunks))]

    def local_result(self):
        """Run this rank's slices group by group and combine them (split format); an exact zero
        when the rank owns none."""
        if self.bc is None:
            return np.zeros(self.out_shape, dtype=self.np_dtype), np.zeros(())
        logs = np.zeros(len(self.my_slices))
        for c0, n, launch in self._chunks:
            launch()
            logs[c0:c0 + n] = self.bc.fetch_log_scale()[:n]   # waits for the group
        t = self.out.cpu().numpy()
        self.last_slices = (t, logs)
        return combine_split([(t[r], logs[r]) for r in range(len(self.my_slices))])

    def run(self, group=None):
        """Contract all slices of this rank and join the ranks' partial results.

        A small result (a closed network's amplitude, a few thousand elements) is joined ON THE DEVICE
        (:meth:`run_small`): the slices' ``(T_hat_s, c_s)`` never leave it, one combine kernel, ONE ``all_gather``
        of the packed ``(T_hat, c)`` - latency-bound, no ring - one more combine, one copy to the host.  A large
        open result takes :meth:`run_device`: reduce-scatter + all-gather over the links, SURVEY.md 8e."""
        if int(np.prod(self.out_shape)) >= DEVICE_JOIN_MIN_NUMEL:
            t, c = self.run_device(group=group)
            return t.cpu().numpy(), np.asarray(c, dtype=np.float64)
        if self.n_total < self.world or os.environ.get("CTN_HOST_JOIN") == "1":
            # more ranks than slices (some rank has no executor to run the combine on), or the development switch:
            # the host join - every rank must take the same route through the collective
            t_loc, c_loc = self.local_result()
            if self.world == 1:
                return t_loc, c_loc
            return all_gather_combine(t_loc, c_loc, group=group, world=self.world, device=self.device)
        return self.run_small(group=group)

    # -- small results: the join stays on the device -------------------------------------------------------------
    def _join_buffers(self):
        import torch

        if self._join is None:
            dev = torch.device("cuda", self.device)
            numel = max(1, int(np.prod(self.out_shape)))
            n_steps = self.bc.plan.n_steps
            self._join = {
                "numel": numel,
                "c": torch.zeros(len(self.my_slices), dtype=torch.float64, device=dev),
                "packed": torch.zeros(numel + 1, dtype=torch.float64, device=dev),
                "gathered": torch.zeros(self.world * (numel + 1), dtype=torch.float64, device=dev),
                "final": torch.zeros(numel + 1, dtype=torch.float64, device=dev),
                # per-step rescale factors of every group of slices, brought over asynchronously: the range check
                # of the lazy rescale (ctn_exec_scales_suspect) costs no extra wait
                "resc": [torch.empty(max(n, 1) * n_steps, dtype=torch.float64).pin_memory() for _c0, n, _l in self._chunks],
            }
            torch.cuda.current_stream(dev).synchronize()     # the zero fills ran on torch's stream
        return self._join

    def run_small(self, group=None):
        """The sliced contraction with a device-resident join; returns ``(T_hat [numpy], c)`` on every rank.

        Per rank: the groups of slices are enqueued back to back; after each one the log-scale registers of its
        slices are copied next to the others' ON the device (and the per-step rescale factors to pinned host memory,
        asynchronously); `k_combine_split` then forms this rank's ``(T_hat_g, c_g) = sum_s T_hat_s e^{c_s}`` in one
        launch, packed as ``numel + 1`` doubles.  Across ranks: ONE ``all_gather`` of that packed buffer (16 bytes
        for a closed network), the same kernel over the ``world`` parts, ONE copy to the host."""
        import torch
        import torch.distributed as dist

        J = self._join_buffers()
        ex, numel = self.bc.executor, J["numel"]
        dev = torch.device("cuda", self.device)
        with ex.lock:
            for k, (c0, n, launch) in enumerate(self._chunks):
                launch()
                ex.snapshot_scales(J["c"].data_ptr() + 8 * c0, n, J["resc"][k].data_ptr())
            ex.combine_split(self.out.data_ptr(), numel, J["c"].data_ptr(), 1, len(self.my_slices), numel,
                             J["packed"].data_ptr())
            ex.synchronize()
            if any(ex.scales_suspect(J["resc"][k].data_ptr(), n) for k, (_c0, n, _l) in enumerate(self._chunks)):
                # a lazily rescaled product left the dtype's range: this rank's part again through the checked path
                # (ctn_exec_fetch repeats such a group with eager rescaling)
                t_loc, c_loc = self.local_result()
                host = np.concatenate([np.asarray(t_loc, dtype=np.float64).ravel(), [float(c_loc)]])
                J["packed"].copy_(torch.from_numpy(host))
                torch.cuda.current_stream(dev).synchronize()
            host = join_packed(ex, J, self.world, group, dev)
        t = host[:numel].reshape(self.out_shape).astype(self.np_dtype)
        return t, np.asarray(host[numel], dtype=np.float64)

    # -- large open outputs: everything stays on the device -------------------------------------------------------
    def local_result_device(self):
        """This rank's slices combined ON THE DEVICE: ``(T_hat_loc [torch, out_shape], c_loc | None)``.

        The slices' results ``(T_hat_s, c_s)`` sit in one device tensor ``out[s, ...]``; their split-format sum
        ``sum_s T_hat_s e^{c_s}`` is itself a contraction - weights ``w_s = e^{c_s - c*}`` against ``out`` over the
        slice label - so it runs on the engine like any other step and comes back stabilised.  ``None`` for the
        scale means "this rank contributes nothing" (no slices, or all of them exact zeros)."""
        import torch

        from .einsum import contract

        dev = torch.device("cuda", self.device)
        if self.bc is None:
            return torch.zeros(self.out_shape, device=dev, dtype=self.out_dtype_torch()), None
        logs = np.zeros(len(self.my_slices))
        for c0, n, launch in self._chunks:
            launch()
            _dev_log, resc = self.bc.executor.fetch()            # waits for the group
            from .einsum import accumulate_log_scale

            for r in range(n):
                logs[c0 + r] = accumulate_log_scale(resc[r], self.np_dtype)
        # a slice counts unless its result is an EXACT zero - the rule of `combine_split`; a tiny result that was
        # never rescaled (sum |T| <= 1e-7) still contributes, weighted like the others
        flat = self.out.reshape(len(self.my_slices), -1)
        live = (flat != 0).any(dim=1).cpu().numpy()
        if not live.any():
            return torch.zeros(self.out_shape, device=dev, dtype=self.out_dtype_torch()), None
        c_star = float(np.max(logs[live]))
        w = np.where(live, np.exp(logs - c_star), 0.0).astype(self.np_dtype)
        w_dev = torch.as_tensor(w, device=dev)
        subs = "".join(chr(ord("a") + i) if i < 25 else chr(ord("A") + i - 25) for i in range(len(self.out_shape)))
        t_loc, c_prime = contract(f"z,z{subs}->{subs}", w_dev, self.out, split_format=True)
        return t_loc, c_star + float(c_prime)

    def out_dtype_torch(self):
        import torch

        return torch.float32 if self.np_dtype == np.float32 else torch.float64

    def run_device(self, group=None):
        """Sliced contraction with a device-resident join; returns ``(T_hat [torch tensor on this rank's GPU], c)``.

        Per rank: :meth:`local_result_device`.  Across ranks (large ``T_hat``): the scales cross
        first (one 8-byte all_gather), every rank brings its tensor to the common scale ``c* = max c_g``, the sum
        is a reduce-scatter (each rank receives and owns 1 / world of the result), the stabilising abs-sum is taken
        on the shards (one scalar all_reduce) and the normalised shards are all-gathered: the tensor crosses the
        links twice, in pieces that use every link, instead of world times through one rank."""
        import torch
        import torch.distributed as dist

        t_loc, c_loc = self.local_result_device()
        if self.world == 1 and not join_alone():
            return t_loc, (0.0 if c_loc is None else c_loc)
        backend = dist.get_backend(group)
        on_dev = backend == "nccl"
        dev = t_loc.device
        cs = torch.tensor([float("-inf") if c_loc is None else c_loc], dtype=torch.float64, device=dev if on_dev else "cpu")
        all_c = [torch.empty_like(cs) for _ in range(self.world)]
        dist.all_gather(all_c, cs, group=group)
        c_star = max(float(x.item()) for x in all_c)
        if c_star == float("-inf"):
            return t_loc, 0.0                                     # every rank holds an exact zero
        u = t_loc.reshape(-1) * (0.0 if c_loc is None else float(np.exp(c_loc - c_star)))
        numel = u.numel()
        if on_dev and numel % self.world == 0:
            shard = torch.empty(numel // self.world, device=dev, dtype=u.dtype)
            dist.reduce_scatter_tensor(shard, u, op=dist.ReduceOp.SUM, group=group)
            norm = shard.abs().sum(dtype=torch.float64).reshape(1)
            dist.all_reduce(norm, op=dist.ReduceOp.SUM, group=group)
            rescale = float(norm.item()) / numel
            if float(norm.item()) > 1e-7:
                shard = shard / rescale
            full = torch.empty(numel, device=dev, dtype=u.dtype)
            dist.all_gather_into_tensor(full, shard, group=group)
        else:   # gloo (CPU tests, one-GPU rehearsals) or a size the shards do not divide: a plain all_reduce
            buf = u if on_dev else u.cpu()
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
            full = buf.to(dev)
            norm_v = float(full.abs().sum(dtype=torch.float64).item())
            rescale = norm_v / numel
            if norm_v > 1e-7:
                full = full / rescale
            norm = torch.tensor([norm_v])
        c = c_star + (float(np.log(rescale)) if float(norm.item()) > 1e-7 else 0.0)
        return full.reshape(self.out_shape), c


def join_alone():
    """``CTN_JOIN_WORLD1=1``: a single rank whose process group exists still issues the collectives of the joins
    (`join_packed`'s all_gather, `run_device`'s reduce-scatter + all_reduce + all-gather) - on a one-GPU box this is how
    the RCCL path is executed on hardware at all (tests/test_gpu_dist.py, `bench.py` under a one-rank launcher)."""
    if os.environ.get("CTN_JOIN_WORLD1") != "1":
        return False
    import torch.distributed as dist

    return dist.is_available() and dist.is_initialized()


def join_packed(ex, J, world, group, dev):
    """The cross-rank half of the device-side join: ``J["packed"]`` (this rank's ``numel + 1`` doubles, complete and
    visible: the executor's stream has been waited for) -> ONE ``all_gather`` -> `k_combine_split` over the ``world``
    parts -> the packed result on the host."""
    import torch
    import torch.distributed as dist

    numel = J["numel"]
    result = J["packed"]
    if world > 1 or join_alone():
        if dist.get_backend(group) == "nccl":
            dist.all_gather_into_tensor(J["gathered"], J["packed"], group=group)      # THE join
        else:   # gloo (CPU tests, one-GPU rehearsals): the same buffer crosses through the host
            send = J["packed"].cpu()
            recv = [torch.empty_like(send) for _ in range(world)]
            dist.all_gather(recv, send, group=group)
            J["gathered"].copy_(torch.cat(recv))
        torch.cuda.current_stream(dev).synchronize()
        g = J["gathered"].data_ptr()
        ex.combine_split(g, numel + 1, g + 8 * numel, numel + 1, world, numel, J["final"].data_ptr(), dtype=np.float64)
        ex.synchronize()
        result = J["final"]
    return result.cpu().numpy()


# ---------------------------------------------------------------------------
# staged slicing: slice-independent work is done once
# ---------------------------------------------------------------------------
def stage_decomposition(einstr, shapes, slice_labels, path, min_saved=1 << 28):
    """Cut the contraction tree of ``path`` into STAGES by the sliced labels each node depends on.

    A node of the tree depends on a sliced label when one of the leaves below it carries that label; nodes with the
    same dependency set that hang together form a stage, and a stage is evaluated once per joint value of ITS labels,
    not once per slice: a subtree that touches no sliced label is contracted a single time, one that touches only the
    first label ``extent(first)`` times, and only the top of the tree - the root stage, which depends on all of them -
    once per slice.  Plain slicing repeats everything for every slice (8 x 8 PEPS, D = 8, 64 slices: 2.1 x the unsliced
    work; the same labels and tree in stages: 1.46 x, and 1.25 x when tree and labels are searched for this form,
    `choose_staged_slices`).

    Returns the stages in execution order (operands before their consumer), each a dict::

        dep       labels the stage depends on, in the order of ``slice_labels``
        operands  [("in", i) | ("st", k)]: network input i, or the result of stage k
        einsum    einsum string over those operands with every sliced label removed (it is fixed per evaluation)
        path      linear path over the stage's operands
        out       labels of the stage's result (the network's output for the last, the root stage)

    A subtree is given a stage of its own only when that saves at least ``min_saved`` multiply-adds (its work times
    the evaluations it is spared); below that it stays inside its consumer and is simply recomputed - a stage is a
    plan, an executor and a handful of launches (the 64 physical-leg absorptions of an 8 x 8 PEPS depend on no
    sliced label and are far too small to earn one each; inside their consumer they go out as ONE grouped launch).

    Pure host function; ``path`` indexes the network's operands (a sliced label keeps its place in the operand list:
    the sliced network has the same operands)."""
    shapes = [tuple(int(d) for d in sh) for sh in shapes]
    terms, out, sizes = paths.parse_einsum_input(einstr, shapes)
    sl = list(slice_labels)
    for lab in sl:
        if lab in out or lab not in sizes:
            raise ValueError(f"cannot slice over label '{lab}'")
    n = len(terms)
    sets = [set(t) for t in terms]
    total = {}
    for t in sets:
        for lab in t:
            total[lab] = total.get(lab, 0) + 1
    # the tree: leaves 0..n-1, internal nodes n.. in path order
    kids, count, dep = {}, {i: {lab: 1 for lab in t} for i, t in enumerate(sets)}, {}
    for i, t in enumerate(sets):
        dep[i] = frozenset(lab for lab in sl if lab in t)
    live, nxt = list(range(n)), n
    for step in path:
        step = tuple(sorted(int(p_) for p_ in step))
        if len(step) == 1:
            continue        # (a unary step belongs to its operand; the engine's path search emits none for n > 1)
        if len(step) != 2:
            raise NotImplementedError("stage_decomposition needs a pairwise path")
        a, b = live[step[0]], live[step[1]]
        live = [x for k, x in enumerate(live) if k not in step] + [nxt]
        kids[nxt] = (a, b)
        c = dict(count[a])
        for lab, v in count[b].items():
            c[lab] = c.get(lab, 0) + v
        count[nxt] = c
        dep[nxt] = dep[a] | dep[b]
        nxt += 1
    if len(live) != 1:
        raise ValueError("path does not reduce the network to a single tensor")
    root = live[0]
    out_set = set(out)

    def labels_of(v):
        if v < n:
            return set(sets[v])
        return {lab for lab, c in count[v].items() if lab in out_set or c < total[lab]}

    def result_term(v):
        """Axis order of a stage's result.  The engine lays a step's output out as [rows from the left operand][columns
        from the right one] and honours the caller's order for the LAST step of a plan only - which a stage's result
        is: the labels that come from the second child of the stage's last step go innermost (columns: 16-byte stores,
        the large-tile kernels stay eligible), each side sorted by extent."""
        if v == root:
            return out
        keep = labels_of(v) - set(sl)
        a, b = kids[v]
        from_b = keep & labels_of(b)
        if not from_b or from_b == keep:
            return "".join(lab for _d, lab in sorted((sizes[lab], lab) for lab in keep))
        side_a = [lab for _d, lab in sorted((sizes[lab], lab) for lab in keep - from_b)]
        side_b = [lab for _d, lab in sorted((sizes[lab], lab) for lab in from_b)]
        return "".join(side_a + side_b)

    sz1 = dict(sizes)
    for lab in sl:
        sz1[lab] = 1

    def evaluations(d):
        e = 1
        for lab in d:
            e *= sizes[lab]
        return e

    sub_work = {}

    def work_below(v):     # multiply-adds of ONE evaluation of the whole subtree under v (sliced extents)
        if v < n:
            return 0
        if v not in sub_work:
            a, b = kids[v]
            sub_work[v] = work_below(a) + work_below(b) + paths._size(labels_of(a) | labels_of(b), sz1)
        return sub_work[v]

    stages, stage_of = [], {}

    def build(r):
        """The stage rooted at internal node r (its operands' stages first)."""
        operands, op_terms, pairs = [], [], []
        ids = {}

        def walk(v):
            if v < n:
                ids[v] = len(operands)
                operands.append(("in", v))
                op_terms.append("".join(lab for lab in terms[v] if lab not in sl))
                return
            if v != r and dep[v] != dep[r] and work_below(v) * (evaluations(dep[r]) - evaluations(dep[v])) >= min_saved:
                if v not in stage_of:
                    build(v)
                ids[v] = len(operands)
                operands.append(("st", stage_of[v]))
                op_terms.append(stages[stage_of[v]]["out"])
                return
            for c in kids[v]:
                walk(c)

        walk(r)
        n_ops = len(operands)
        steps = []

        def emit(v):
            if v in ids:
                return ids[v]
            a, b = (emit(c) for c in kids[v])
            ids[v] = n_ops + len(steps)
            steps.append((a, b))
            return ids[v]

        emit(r)
        res = result_term(r)
        stage_of[r] = len(stages)
        stages.append({"dep": tuple(lab for lab in sl if lab in dep[r]), "operands": operands,
                       "einsum": ",".join(op_terms) + "->" + res, "path": tuple(paths._ssa_pairs_to_linear(steps, n_ops)),
                       "out": res})

    if root < n:
        raise ValueError("a single-operand network has nothing to stage")
    build(root)
    return stages


def choose_staged_slices(einstr, shapes, min_slices=1, max_intermediate=None, parallel=8, max_held=1 << 34, max_labels=6,
                         seeds=4, rounds=8, first_seed=0):
    """Slice labels and contraction tree for STAGED execution (`stage_decomposition`), chosen together.

    Objective: ``paths.hoisted_cost(..., parallel)`` - every node of the tree costs its sliced index space times the
    number of joint values of the sliced labels BELOW it, and at least ``parallel`` evaluations (work that cannot be
    dealt to ``parallel`` ranks is replicated on them) - subject to at least ``min_slices`` slices, no tensor above
    ``max_intermediate`` elements and at most ``max_held`` elements held between stages.  Search, per seed: a tree
    (the library's own path for seed 0, refined noisy-greedy trees for the others), then alternately (a) the label set
    on that tree - greedy by the objective, then swaps and removals until none helps - and (b) subtree
    reconfiguration of the tree for those labels under the same objective (`paths._reconfigure(slice_labels=...)`),
    until neither moves.  The best seed wins (``first_seed``: where the seed range starts, for searches spread over
    processes - tools/make_plans.py).  Returns ``(labels, path, report)`` like `choose_slices_with_path`;
    ``report["work_overhead"]`` is total work over all evaluations / unsliced work, ``report["plain_overhead"]`` what
    the same labels and tree cost when every slice repeats everything.  When materialising every stage for all its
    evaluations would hold more than ``max_held`` elements (8 x 8 PEPS, D = 16: stage results of 2^28 elements each),
    the first ``report["outer_labels"]`` of the returned labels are meant to be walked in a host loop, one value at
    a time (`StagedSlicedContraction(outer=...)`), and the label ORDER is chosen with that count so that what has to be
    recomputed per outer value costs least (`paths.hoisted_profile`).  Host-only, deterministic."""
    shapes = [tuple(int(d) for d in sh) for sh in shapes]
    terms, out, sizes = paths.parse_einsum_input(einstr, shapes)
    sets = [set(t) for t in terms]
    cands = sorted(lab for lab in sizes if lab not in out and sizes[lab] > 1)

    plan_of = {}     # (labels in search order) -> (labels in loop order, outer) of the last measurement

    def measure(path, labels):
        """Objective and constraints for this label SET: everything materialised at once when that fits ``max_held``,
        otherwise the cheapest (loop order, number of outer labels walked on the host) that does."""
        cnt = 1
        for lab in labels:
            cnt *= sizes[lab]
        tot, big, _one, held = paths.hoisted_profile(sets, out, sizes, path, labels, parallel=parallel)
        plan_of[tuple(labels)] = (tuple(labels), 0)
        if held > max_held and 0 < len(labels) <= 4:
            best_ = None
            for order in itertools.permutations(labels):
                for outer in range(1, len(labels) + 1):
                    t2, b2, _o, h2 = paths.hoisted_profile(sets, out, sizes, path, order, parallel=parallel, outer=outer)
                    key = (h2 > max_held, t2 if h2 <= max_held else h2, outer, order)
                    if best_ is None or key < best_[0]:
                        best_ = (key, t2, b2, h2, order, outer)
            _k, tot, big, held, order, outer = best_
            plan_of[tuple(labels)] = (tuple(order), outer)
        return tot, big, cnt, held

    def feasible(big, cnt, held):
        return cnt >= min_slices and (max_intermediate is None or big <= max_intermediate) and held <= max_held

    def violation(big, cnt, held):
        v = 0.0
        if max_intermediate is not None and big > max_intermediate:
            v += big / max_intermediate
        if held > max_held:
            v += held / max_held
        return v

    def pick_labels(path, start):
        labels = list(start)
        tot, big, cnt, held = measure(path, labels)
        while not feasible(big, cnt, held) and len(labels) < max_labels:
            best = None
            for lab in cands:
                if lab in labels:
                    continue
                t2, b2, c2, h2 = measure(path, labels + [lab])
                key = (violation(b2, c2, h2), t2, lab)
                if best is None or key < best[0]:
                    best = (key, lab)
            if best is None:
                break
            labels.append(best[1])
            tot, big, cnt, held = measure(path, labels)
        moved = feasible(big, cnt, held)
        while moved:
            moved = False
            for i in range(len(labels)):
                for lab in cands:
                    if lab in labels:
                        continue
                    trial = labels[:i] + [lab] + labels[i + 1:]
                    t2, b2, c2, h2 = measure(path, trial)
                    if feasible(b2, c2, h2) and t2 < tot:
                        labels, tot, big, cnt, held, moved = trial, t2, b2, c2, h2, True
            for i in range(len(labels)):
                trial = labels[:i] + labels[i + 1:]
                t2, b2, c2, h2 = measure(path, trial)
                if feasible(b2, c2, h2) and t2 < tot:
                    labels, tot, big, cnt, held, moved = trial, t2, b2, c2, h2, True
                    break
        return labels, tot, big, cnt, held

    base_path = paths.find_path(terms, out, sizes, "auto")
    base_flops, base_big = paths.path_cost(sets, out, sizes, base_path)
    best = None
    for seed in range(first_seed, first_seed + max(1, seeds)):
        if seed == 0:
            path = base_path
        else:
            path = paths._reconfigure(sets, out, sizes, paths._random_greedy(sets, out, sizes, repeats=4, seed=seed),
                                      max_leaves=8, rounds=8)
        labels = []
        for _ in range(rounds):
            labels, tot, big, cnt, held = pick_labels(path, labels)
            if feasible(big, cnt, held) and (best is None or (tot, big) < best[0]):
                best = ((tot, big), tuple(labels), path, cnt, held)
            path2 = paths._reconfigure(sets, out, sizes, path, max_leaves=8, rounds=8, memory_limit=max_intermediate,
                                       slice_labels=labels, parallel=parallel)
            t2, b2, c2, h2 = measure(path2, labels)
            if feasible(b2, c2, h2) and (best is None or (t2, b2) < best[0]):
                best = ((t2, b2), tuple(labels), path2, c2, h2)
            if t2 >= tot:
                break
            path = path2
    if best is None:
        raise ValueError("no feasible staged slicing found: relax max_intermediate / max_held or allow more labels")
    (tot, big), labels, path, cnt, held = best
    measure(path, list(labels))
    labels, outer = plan_of[tuple(labels)]      # loop order: the first `outer` labels are walked on the host
    work, _b, one, held = paths.hoisted_profile(sets, out, sizes, path, labels, parallel=1, outer=outer)
    report = {"slices": cnt, "largest_intermediate": big, "unsliced_largest_intermediate": base_big,
              "work_overhead": work / max(base_flops, 1), "plain_overhead": one * cnt / max(base_flops, 1),
              "modelled_overhead_at_parallel": tot / max(base_flops, 1), "parallel": parallel, "outer_labels": outer,
              "held_between_stages": held, "unsliced_flops": base_flops, "staged": True}
    return tuple(labels), tuple(tuple(int(x) for x in p_) for p_ in path), report


def staged_plan(einstr, shapes, min_slices=1, max_intermediate=None, cache_dir=None, **kwargs):
    """`choose_staged_slices` behind the same file cache as `sliced_plan` (``staged_<hash>.json``)."""
    import hashlib
    import json

    shapes = [tuple(int(d) for d in sh) for sh in shapes]
    cache_dir = cache_dir or os.path.join(os.path.dirname(os.path.abspath(__file__)), "plans")
    key = hashlib.sha1(json.dumps(["staged", einstr, shapes, int(min_slices), max_intermediate, sorted(kwargs.items())],
                                  ensure_ascii=True).encode()).hexdigest()[:16]
    fname = os.path.join(cache_dir, f"staged_{key}.json")
    if os.path.exists(fname):
        try:
            with open(fname) as fh:
                d = json.load(fh)
            labels = tuple(d["labels"])
            path = tuple(tuple(int(x) for x in p_) for p_ in d["path"])
            lhs, out = einstr.split("->")
            if (d.get("einsum_str") == einstr and all(lab in lhs and lab not in out for lab in labels)
                    and len(path) == len(shapes) - 1):
                return labels, path, d["report"]
        except (OSError, ValueError, KeyError):
            pass
    labels, path, report = choose_staged_slices(einstr, shapes, min_slices=min_slices, max_intermediate=max_intermediate,
                                                **kwargs)
    try:
        os.makedirs(cache_dir, exist_ok=True)
        with open(fname, "w") as fh:
            json.dump({"einsum_str": einstr, "shapes": shapes, "min_slices": int(min_slices),
                       "max_intermediate": max_intermediate, "labels": list(labels),
                       "path": [list(p_) for p_ in path], "report": report}, fh)
    except OSError:
        pass
    return labels, path, report



def _best_rank_grid(extents, world, dep_idx, stage_work):
    """Factor ``world`` ranks over the sliced labels (``grid[i]`` ranks along label i, every factor at most the label's
    extent) so that the work of the busiest rank is least: a stage costs a rank its work per evaluation times the
    number of joint values of the stage's labels inside the rank's block."""
    n = len(extents)
    best = None

    def blocks(e_, g_):
        return -(-e_ // g_)          # the largest share of a range of e_ cut into g_ parts

    def rec(i, left, grid):
        nonlocal best
        if i == n:
            if left != 1:
                return
            cost = sum(w_ * int(np.prod([blocks(extents[j], grid[j]) for j in dep])) for dep, w_ in zip(dep_idx, stage_work))
            key = (cost, tuple(-g_ for g_ in grid))      # ties: ranks along the leading labels first
            if best is None or key < best[0]:
                best = (key, tuple(grid))
            return
        for g_ in range(1, min(left, extents[i]) + 1):
            if left % g_ == 0:
                rec(i + 1, left // g_, grid + [g_])

    rec(0, world, [])
    if best is None:
        raise ValueError(f"{world} ranks cannot be factored over sliced labels of extents {extents}")
    return best[1]


class StagedSlicedContraction:
    """Index slicing in STAGES (`stage_decomposition`): every stage is one plan whose evaluations - one per joint
    value of the sliced labels it depends on - run as replicas, lower stages first; their results stay on the device
    as operands of the stages above (an evaluation is a pointer offset, like an input's slice), only the root stage
    runs once per slice.  Everything is enqueued on ONE stream with no host wait in between; the scales of lower
    stages are added to their consumers' on the device, the root's ``(T_hat_s, c_s)`` are summed by
    `k_combine_split` and joined across ranks exactly like `SlicedContraction.run_small` (ONE ``all_gather``).

    ``outer``: the first ``outer`` labels (in the order given) are walked in a host loop, one joint value - one GROUP
    of slices - at a time, so that a stage only ever holds the evaluations one group needs (8 x 8 PEPS, D = 16: stage
    results of 2^28 elements; all 256 evaluations of one would be 256 GiB).  A stage is recomputed for a group only
    when the evaluations that group needs differ from the ones it holds (`paths.hoisted_profile(outer=...)` counts
    exactly that).  ``None`` = the smallest count whose buffers fit ``held_budget`` bytes.

    ``unslice``: at the ROOT stage a sliced label may become an ordinary contracted label again - its values then lie
    side by side in the stage buffers below (one more, strided axis of those operands) and the root sums over it
    inside its own GEMMs instead of once per value.  Worth it when another large operand of the root does not depend
    on the label: 8 x 8 PEPS, D = 16, the root's 2^28-element operand was read 16 times, once per value of the third
    label, by GEMMs with 16 rows; un-sliced, it is read once by one GEMM with 256 rows.  The slices of such an operand
    come with different scales: they are brought to their common maximum first (one pass over the buffer, on the
    device).  Only labels that are not walked on the host, while the root's intermediates stay below ``unslice_limit``.

    Ranks: the grid of label values is cut into one BLOCK per rank (`_best_rank_grid`: the factorisation of the
    ranks over the labels under which the busiest rank has least to do); of a lower stage a rank evaluates what its
    block projects onto (a stage that depends on no sliced label is computed by every rank - that work does not
    scale, which is what ``parallel`` in `choose_staged_slices` prices).  Small results only (closed networks, a few thousand elements);
    ``optimize`` must be an explicit linear path (the one the labels were chosen with)."""

    def __init__(self, einstr, operands, slice_labels, optimize, rank=0, world=1, device=0, dtype=None,
                 workspace_budget=64 << 30, min_saved=1 << 28, outer=None, held_budget=64 << 30, unslice=True,
                 unslice_limit=1 << 28, unslice_min_numel=1 << 22):
        import torch

        from . import einsum as E

        if isinstance(optimize, (str, bool)) or optimize is None:
            raise ValueError("StagedSlicedContraction needs the explicit path its labels were chosen with")
        self.rank, self.world, self.device = rank, world, device
        self._args = (einstr, operands, tuple(slice_labels), _hashable(optimize), dtype, workspace_budget)
        shapes = [tuple(np.shape(o)) for o in operands]
        terms, out, sizes = paths.parse_einsum_input(einstr, shapes)
        self.sizes, self.slice_labels = sizes, tuple(slice_labels)
        self.np_dtype = np.dtype(dtype or np.result_type(*[np.asarray(o).dtype for o in operands]))
        if self.np_dtype not in (np.float32, np.float64):
            self.np_dtype = np.dtype(np.float64)
        tdt = torch.float32 if self.np_dtype == np.float32 else torch.float64
        item = self.np_dtype.itemsize
        dev = torch.device("cuda", device)
        self.out_shape = tuple(sizes[c] for c in out)
        if int(np.prod(self.out_shape)) >= DEVICE_JOIN_MIN_NUMEL:
            raise NotImplementedError("staged slicing joins small results only; use SlicedContraction for large open outputs")
        self.stage_desc = stage_decomposition(einstr, shapes, self.slice_labels, optimize, min_saved=min_saved)
        at = {lab: i for i, lab in enumerate(self.slice_labels)}
        n_stage = len(self.stage_desc)
        dep_idx = [[at[lab] for lab in st["dep"]] for st in self.stage_desc]
        extents = [sizes[lab] for lab in self.slice_labels]
        self.n_total = int(np.prod(extents)) if extents else 1
        if self.n_total < world:
            raise ValueError(f"{self.n_total} slices cannot be dealt to {world} ranks")
        # which slices are this rank's: a BLOCK of the grid of label values - ranks factored over the labels, a
        # contiguous share of every label's range - chosen so that the stages below the root are evaluated as few times
        # as possible (a rank evaluates a stage once per value of the stage's labels that its slices touch: a 2 x 4
        # block of an 8 x 8 grid touches 2 + 4 values of the two labels, a 1 x 8 row 1 + 8)
        stage_work = []
        for st in self.stage_desc:
            lhs_ = st["einsum"].split("->")[0].split(",")
            stage_work.append(paths.path_cost([set(t_) for t_ in lhs_], st["out"], sizes, st["path"])[0] if len(lhs_) > 1 else 0)
        self.rank_grid = _best_rank_grid(extents, world, dep_idx, stage_work)
        coords, rem = [], rank
        for g_ in reversed(self.rank_grid):
            coords.append(rem % g_)
            rem //= g_
        coords.reverse()
        self.my_slices = list(itertools.product(*[shard_range(e_, c_, g_) for e_, c_, g_ in zip(extents, coords, self.rank_grid)]))
        stage_numel, stage_stride = [], []
        for st in self.stage_desc:
            numel = int(np.prod([sizes[c] for c in st["out"]])) if st["out"] else 1
            stage_numel.append(numel)
            stage_stride.append((numel * item + 15) // 16 * 16 // item)     # evaluations start on 16-byte boundaries

        def grouping(n_outer):
            """Groups of this rank's slices by the values of the first n_outer labels, and per stage and group the
            evaluations (tuples over the stage's own labels) the group needs, in order."""
            groups = {}
            for v in self.my_slices:
                groups.setdefault(v[:n_outer], []).append(v)
            order = list(groups)
            need = [[sorted({tuple(v[i] for i in dep_idx[k]) for v in groups[g]}) for g in order] for k in range(n_stage)]
            return [groups[g] for g in order], need

        if outer is None:      # the fewest outer labels whose stage buffers fit the budget
            for outer in range(len(self.slice_labels) + 1):
                _g, need = grouping(outer)
                held = sum(max(len(x) for x in need[k]) * stage_stride[k] * item for k in range(n_stage - 1))
                if held <= held_budget:
                    break
        self.outer = int(outer)
        groups, need = grouping(self.outer)
        self.n_groups = len(groups)
        # labels the root takes back as ordinary contracted labels (see the class docstring)
        root_desc = self.stage_desc[-1]
        self.unsliced = ()
        if unslice and n_stage > 1:
            chosen = []
            for lab in reversed(self.slice_labels[self.outer:]):          # innermost (fastest-varying) label first
                carriers, others = [], []
                for kind, j in root_desc["operands"]:
                    has = (lab in terms[j]) if kind == "in" else (lab in self.stage_desc[j]["dep"])
                    (carriers if has else others).append((kind, j))
                if not carriers or any(kind == "in" for kind, _j in carriers):
                    continue
                if not any(kind == "st" and stage_numel[j] >= unslice_min_numel for kind, j in others):
                    continue                                              # nothing large is re-read: not worth a pass
                trial = chosen + [lab]
                r_terms = []
                for kind, j in root_desc["operands"]:
                    if kind == "in":
                        r_terms.append("".join(c for c in terms[j] if c not in self.slice_labels))
                    else:
                        r_terms.append("".join(c for c in self.stage_desc[j]["dep"] if c in trial) + self.stage_desc[j]["out"])
                big = paths.path_cost([set(t_) for t_ in r_terms], out, sizes, root_desc["path"])[1] if len(r_terms) > 1 else 0
                if big <= unslice_limit:
                    chosen = trial
            self.unsliced = tuple(lab for lab in self.slice_labels if lab in chosen)
        root_dep = [lab for lab in root_desc["dep"] if lab not in self.unsliced]
        dep_idx[-1] = [at[lab] for lab in root_dep]
        need[-1] = [sorted({tuple(v[i] for i in dep_idx[-1]) for v in grp}) for grp in groups]
        self.root_members = []      # per root evaluation (in buffer order): indices into my_slices of the slices it sums
        where = {v: q for q, v in enumerate(self.my_slices)}
        for grp, wanted in zip(groups, need[-1]):
            for e in wanted:
                self.root_members.append([where[v] for v in grp if tuple(v[i] for i in dep_idx[-1]) == e])

        self.tstream = torch.cuda.Stream(dev)
        # Stages that do not feed each other (the quadrants of a 2D grid) are chains of small, latency-bound launches:
        # each gets a stream of its own (four in rotation; the root keeps the main one) and events carry the tree's
        # dependencies, so that one stage's launch latencies hide behind another's kernels.  Only when every stage is
        # evaluated once per contraction (one group; the host loop over `outer` labels re-uses buffers between groups).
        multi = n_stage > 2 and len(groups) == 1 and os.environ.get("CTN_STAGE_STREAMS", "1") != "0"
        pool = [torch.cuda.Stream(dev) for _ in range(min(4, n_stage - 1))] if multi else []
        self.stage_streams = [pool[k % len(pool)] if pool and k < n_stage - 1 else self.tstream for k in range(n_stage)]
        # inputs: sliced axes first (in slice_labels order), so a slice is one contiguous block at a pointer offset
        self.tensors, lead_labels, blocks = [], [], []
        for term, op in zip(terms, operands):
            lead = sorted((i for i, c in enumerate(term) if c in self.slice_labels), key=lambda i: at[term[i]])
            rest = [i for i, c in enumerate(term) if c not in self.slice_labels]
            t = torch.as_tensor(np.ascontiguousarray(np.transpose(np.asarray(op, dtype=self.np_dtype), lead + rest)),
                                device=dev, dtype=tdt)
            self.tensors.append(t)
            lead_labels.append(tuple(term[i] for i in lead))
            blocks.append(int(np.prod([sizes[term[i]] for i in rest])) if rest else 1)
        self._owned = {}

        def input_ptr(i, values):
            idx = 0
            for lab in lead_labels[i]:
                idx = idx * sizes[lab] + values[at[lab]]
            p_ = self.tensors[i].data_ptr() + idx * blocks[i] * item
            if p_ % 16:      # odd-sized block: an aligned copy of this slice
                if (i, idx) not in self._owned:
                    self._owned[(i, idx)] = self.tensors[i][tuple(values[at[lab]] for lab in lead_labels[i])].clone()
                p_ = self._owned[(i, idx)].data_ptr()
            return p_

        # per stage: plan, executor, result buffer (as many evaluations as the largest group needs; the root: one per slice)
        self.stages = []
        n_local = len(self.my_slices)
        for k, st in enumerate(self.stage_desc):
            root = k == n_stage - 1
            n_buf = len(self.root_members) if root else max(len(x) for x in need[k])
            # the stage's einsum over its operands: a lower stage's result in the axis order its plan chose (below the
            # root the engine lays results out itself - rows of the last step's left operand, then columns of its right
            # one - so that the step keeps its 16-byte stores and large tiles; only the root has the caller's order)
            lhs, st_shapes, st_strides = [], [], []
            for kind, j in st["operands"]:
                if kind == "in":
                    t_ = "".join(c for c in terms[j] if c not in self.slice_labels)
                    lhs.append(t_)
                    st_shapes.append(tuple(sizes[c] for c in t_))
                    st_strides.append(None)
                    continue
                child = self.stages[j]
                inner = tuple(sizes[c] for c in child["out_term"])
                inner_strides = [1] * len(inner)
                for a_ in range(len(inner) - 2, -1, -1):
                    inner_strides[a_] = inner_strides[a_ + 1] * inner[a_ + 1]
                extra = [lab for lab in child["desc"]["dep"] if root and lab in self.unsliced]
                # an un-sliced label is one more axis of the child's buffer: its values lie `stride x (evaluations per
                # step of the label)` elements apart (the evaluations of a group are a grid over the child's labels)
                ex_shape, ex_stride = [], []
                for lab in extra:
                    pos_ = child["desc"]["dep"].index(lab)
                    counts = [len({e[q] for e in need[j][0]}) for q in range(len(child["desc"]["dep"]))]
                    ex_shape.append(counts[pos_])
                    ex_stride.append(int(np.prod(counts[pos_ + 1:])) * child["stride"])
                lhs.append("".join(extra) + child["out_term"])
                st_shapes.append(tuple(ex_shape) + inner)
                st_strides.append(tuple(ex_stride) + tuple(inner_strides) if extra else None)
            st_einsum = ",".join(lhs) + "->" + st["out"]
            have_strides = any(x is not None for x in st_strides)
            if have_strides:
                def contiguous(shape):
                    out_, acc = [], 1
                    for d_ in reversed(shape):
                        out_.append(acc)
                        acc *= d_
                    return tuple(reversed(out_))
                st_strides = tuple(x if x is not None else contiguous(shp) for x, shp in zip(st_strides, st_shapes))
            else:
                st_strides = None
            probe = E._native_plan(E._contract_path(st_einsum, tuple(st_shapes), optimize=_hashable(st["path"]),
                                                    memory_limit=None, use_blas=True), tuple(st_shapes), self.np_dtype.name,
                                   not root, st_strides)
            per = max(1, probe.workspace_bytes(2) - probe.workspace_bytes(1))
            R = int(max(1, min(max(len(x) for x in need[k]), workspace_budget // per)))
            # ... and no more evaluations in flight than the card has room for RIGHT NOW (the budget is per stage; the
            # stages below, their result buffers and whatever else the process holds are already allocated), 8 GiB spared
            try:
                free_b = torch.cuda.mem_get_info(dev)[0]
                own = stage_stride[k] * item
                R = int(max(1, min(R, (free_b - (8 << 30) - n_buf * own) // (per + own))))
            except (RuntimeError, AssertionError, TypeError):
                pass
            bc = E.BatchedContraction(st_einsum, st_shapes, self.np_dtype, optimize=st["path"], replicas=R, device=device,
                                      stream=self.stage_streams[k].cuda_stream, free_output_order=not root, in_strides=st_strides)
            self.stages.append({
                "desc": st, "bc": bc, "R": R, "numel": stage_numel[k], "stride": stage_stride[k],
                "out_term": st["out"] if root else bc.out_subscripts,
                "out": torch.zeros((n_buf, stage_stride[k]), device=dev, dtype=tdt),
                "scratch": torch.zeros((R, stage_stride[k]), device=dev, dtype=tdt),
                "c": torch.zeros(n_buf, dtype=torch.float64, device=dev),
                "cum": torch.zeros(n_buf, dtype=torch.float64, device=dev),
                "launches": 0, "evaluated": 0})

        # the schedule: group by group, every stage whose needed evaluations changed is recomputed into its buffer
        self.schedule = []
        holds = [None] * n_stage                     # evaluations currently in each stage's buffer, in slot order
        done = 0                                     # root slices scheduled so far (= their slot in the root buffer)
        for g, slices in enumerate(groups):
            for k, st in enumerate(self.stage_desc):
                S = self.stages[k]
                root = k == n_stage - 1
                want = need[k][g]
                if not root and holds[k] == want:
                    continue
                holds[k] = want
                slot0 = done if root else 0
                pos_child = {j: {e: q for q, e in enumerate(holds[j])} for kind, j in st["operands"] if kind == "st"}
                # the child's evaluation for one of ours: our label values, and for a label the root has un-sliced the
                # FIRST value of the group (the others lie behind it along the extra axis)
                first = {lab: min(v[at[lab]] for v in slices) for lab in self.unsliced} if root else {}

                def child_eval(e, j, k=k, first=first):
                    mine = dict(zip(dep_idx[k], e))
                    return tuple(mine[at[lab]] if at[lab] in mine else first[lab] for lab in self.stage_desc[j]["dep"])

                if root and self.unsliced:
                    for j in pos_child:                      # bring the slices along the un-sliced axes to a common scale
                        dep_j = self.stage_desc[j]["dep"]
                        axes = [q for q, lab in enumerate(dep_j) if lab in self.unsliced]
                        if axes:
                            grid = [len({e[q] for e in holds[j]}) for q in range(len(dep_j))]
                            self.schedule.append(("merge", j, len(holds[j]), tuple(grid), tuple(axes)))

                def eval_ptrs(e, st=st, k=k, pos_child=pos_child, child_eval=child_eval):
                    full = [0] * len(self.slice_labels)
                    for i, v in zip(dep_idx[k], e):
                        full[i] = v
                    ptrs = []
                    for kind, j in st["operands"]:
                        if kind == "in":
                            ptrs.append(input_ptr(j, full))
                        else:
                            q = pos_child[j][child_eval(e, j)]
                            ptrs.append(self.stages[j]["out"].data_ptr() + q * self.stages[j]["stride"] * item)
                    return ptrs

                R, n_steps = S["R"], S["bc"].plan.n_steps
                for c0 in range(0, len(want), R):
                    chunk = want[c0:c0 + R]
                    n = len(chunk)
                    padded = chunk + [chunk[-1]] * (R - n)
                    in_ptrs = [p_ for e in padded for p_ in eval_ptrs(e)]
                    out_ptrs = [S["out"].data_ptr() + (slot0 + c0 + r) * S["stride"] * item if r < n
                                else S["scratch"].data_ptr() + r * S["stride"] * item for r in range(R)]
                    resc = torch.empty(n * n_steps, dtype=torch.float64).pin_memory()
                    self.schedule.append(("launch", k, S["bc"].executor.make_enqueue(in_ptrs, out_ptrs), slot0 + c0, n, resc))
                    S["launches"] += 1
                    S["evaluated"] += n
                # scales of the stages below ride along with their results: cum = own + sum of the children's
                kids_idx = [(j, torch.as_tensor([pos_child[j][child_eval(e, j)] for e in want], dtype=torch.int64, device=dev))
                            for j in pos_child]
                self.schedule.append(("cum", k, slot0, len(want), kids_idx))
            done += len(need[-1][g])
        root = self.stages[-1]
        numel = max(1, int(np.prod(self.out_shape)))
        assert root["numel"] == numel and done == len(self.root_members)
        self.R = root["R"]
        self._join = {"numel": numel, "packed": torch.zeros(numel + 1, dtype=torch.float64, device=dev),
                      "gathered": torch.zeros(world * (numel + 1), dtype=torch.float64, device=dev),
                      "final": torch.zeros(numel + 1, dtype=torch.float64, device=dev)}
        self._plain = None
        torch.cuda.synchronize(dev)       # uploads, clones and zero fills ran on torch's current stream

    def stage_list(self):
        """``[(BatchedContraction, evaluations on this rank per contraction, replicas per launch, launches)]``."""
        return [(S["bc"], S["evaluated"], S["R"], S["launches"]) for S in self.stages]

    def slices_host(self):
        """``(T_hat [n, ...], c [n])`` of the root stage's evaluations as the LAST run left them on the device (with the
        scales of the stages below added) - for slice-by-slice checks: evaluation q is the sum of this rank's slices
        ``root_members[q]`` (one slice each unless the root took sliced labels back, ``unsliced``)."""
        root = self.stages[-1]
        self.tstream.synchronize()
        t = root["out"][:, :root["numel"]].cpu().numpy().reshape((len(self.root_members),) + self.out_shape)
        return t, root["cum"].cpu().numpy()

    def evaluations(self):
        """Per stage: (labels it depends on, evaluations on this rank per contraction, this rank's slices = what plain
        slicing would evaluate)."""
        return [(S["desc"]["dep"], S["evaluated"], len(self.my_slices)) for S in self.stages]

    def run(self, group=None):
        """One sliced contraction; returns ``(T_hat [numpy], c)`` on every rank."""
        import torch

        dev = torch.device("cuda", self.device)
        J = self._join
        root = self.stages[-1]
        ex = root["bc"].executor
        streams = self.stage_streams
        concurrent = any(st is not self.tstream for st in streams)
        if concurrent:                               # the side streams start behind everything the main one has been given
            start = torch.cuda.Event()
            start.record(self.tstream)
            for st in set(streams):
                if st is not self.tstream:
                    st.wait_event(start)
            done = {}                                # stage -> event behind its last entry
            seen = set()
        for entry in self.schedule:
            k = entry[1]
            if entry[0] == "merge":                  # (works on a child's buffer, on behalf of the root)
                k = len(self.stages) - 1
            stream = streams[k]
            if concurrent and k not in seen:         # first entry of stage k: its operands' stages have finished
                seen.add(k)
                for kind, j in self.stage_desc[k]["operands"]:
                    if kind == "st" and streams[j] is not stream:
                        stream.wait_event(done[j])
            if entry[0] == "launch":
                _tag, k_, launch, slot, n, resc = entry
                S = self.stages[k_]
                launch()
                S["bc"].executor.snapshot_scales(S["c"].data_ptr() + 8 * slot, n, resc.data_ptr())
            elif entry[0] == "merge":
                # the evaluations of stage j along the un-sliced axes become ONE operand of the root: bring them to
                # their common (largest) scale - exact zeros aside - and let that scale ride along (one launch pair on
                # the root's stream: ctn_exec_merge_scales)
                _tag, j, n, grid, axes = entry
                S = self.stages[j]
                ex.merge_scales(S["out"].data_ptr(), S["stride"], S["numel"], S["cum"].data_ptr(), n, grid,
                                [q in axes for q in range(len(grid))])
            else:
                # cum = own register + those of the children's evaluations it consumed (ctn_exec_add_scales, on the
                # stage's own stream)
                _tag, k_, slot0, n, kids_idx = entry
                S = self.stages[k_]
                S["bc"].executor.add_scales(S["cum"].data_ptr() + 8 * slot0, S["c"].data_ptr() + 8 * slot0, n,
                                            [(self.stages[j]["cum"].data_ptr(), idx.data_ptr()) for j, idx in kids_idx])
                if concurrent:                   # ("cum" is a stage's last entry of a group)
                    done[k_] = torch.cuda.Event()
                    done[k_].record(stream)
        ex.combine_split(root["out"].data_ptr(), root["stride"], root["cum"].data_ptr(), 1, len(self.root_members),
                         J["numel"], J["packed"].data_ptr())
        self.tstream.synchronize()
        verdict = [False] * len(self.stages)
        for e in self.schedule:
            if e[0] == "launch" and not verdict[e[1]]:
                verdict[e[1]] = self.stages[e[1]]["bc"].executor.scales_suspect(e[5].data_ptr(), e[4])
        for S, v in zip(self.stages, verdict):           # (a stage that keeps being suspect goes eager for good)
            S["bc"].executor.report_suspect(v)
        suspect = any(verdict)
        if suspect:
            # a lazily rescaled product left the dtype's range somewhere: THIS RANK'S BLOCK of slices again on the plain,
            # checked path (per-slice fetch repeats such a group with eager rescaling; `slices=`: exactly the block the
            # staged form dealt to this rank, whatever the other ranks do); every rank still joins once
            if self._plain is None:
                einstr, operands, labels, path, dtype, budget = self._args
                self._plain = SlicedContraction(einstr, operands, labels, optimize=path, rank=self.rank, world=self.world,
                                                device=self.device, dtype=dtype, workspace_budget=budget,
                                                slices=self.my_slices)
            t_loc, c_loc = self._plain.local_result()
            host = np.concatenate([np.asarray(t_loc, dtype=np.float64).ravel(), [float(c_loc)]])
            J["packed"].copy_(torch.from_numpy(host))
            torch.cuda.current_stream(dev).synchronize()
        host = join_packed(ex, J, self.world, group, dev)
        numel = J["numel"]
        return host[:numel].reshape(self.out_shape).astype(self.np_dtype), np.asarray(host[numel], dtype=np.float64)


DEVICE_JOIN_MIN_NUMEL = 1 << 14     # results at least this large are joined on the device (reduce-scatter + all-gather)
