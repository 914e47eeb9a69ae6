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
* ``"optimal"``: exhaustive depth-first search over pairwise orders (up to 8 operands), ``"dp"``:
  exact minimum-flop order by dynamic programming over operand subsets (up to 12 operands);
* ``"random-greedy[-N]"``: greedy restarts with Boltzmann noise, best of N by flop count;
* ``"auto"``/``True``: optimal below 5 operands, dp up to 12; beyond that 8 random-greedy trials up to 64
  operands, 4 up to 256, plain greedy beyond - refined by subtree reconfiguration (exact DP on every
  subtree of up to 8 branches) unless the greedy path is already flat (opt_einsum: optimal / branch-and-bound / greedy over similar ranges); ``"auto-hq"``:
  128 random-greedy trials at any size, reconfiguration with 10-branch subtrees;
* ``memory_limit`` (elements, or ``"max_input"``): dp and random-greedy prefer paths whose
  intermediates stay below it (they never fail: a larger intermediate is taken if nothing else exists).

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


def path_cost(term_sets, out, sizes, path):
    """(total flop proxy = sum over steps of the size of the joint index space, largest intermediate)."""
    live = [set(t) for t in term_sets]
    flops, biggest = 0, 0
    for step in path:
        step = tuple(sorted(step))
        if len(step) == 1:
            continue
        i, j = step
        rest = [x for k, x in enumerate(live) if k not in (i, j)]
        new, _ = _pair_result(live[i], live[j], rest, out)
        flops += _size(live[i] | live[j], sizes)
        biggest = max(biggest, _size(new, sizes))
        live = rest + [new]
    return flops, biggest


def hoisted_cost(term_sets, out, sizes, path, slice_labels, parallel=1):
    """Work of an index-sliced path when slice-INDEPENDENT work is done once (`dist.StagedSlicing`): a node of the
    tree is evaluated once per joint value of the sliced labels its subtree depends on - those carried by one of its
    leaves - not once per slice.  A label that is contracted at the root therefore slices for free, and a subtree that
    touches no sliced label is computed a single time.  Returns ``(total multiply-adds over all evaluations, largest
    intermediate - sliced where it carries sliced labels -, multiply-adds of one slice evaluated in full)``;
    ``sizes`` are the UNSLICED extents.

    ``parallel`` = the number of ranks the slices are meant to be dealt to: a node evaluated fewer times than that
    cannot be shared out (every rank needs its result: the work is replicated, or the others wait), so it is
    charged ``parallel`` evaluations - the first figure is then ``parallel`` x the modelled time of one rank, and
    a slicing whose sliced labels sit in a corner of the network (nearly everything slice-independent: no overhead,
    no parallelism either) stops looking free."""
    return hoisted_profile(term_sets, out, sizes, path, slice_labels, parallel)[:3]


def hoisted_profile(term_sets, out, sizes, path, slice_labels, parallel=1, outer=0):
    """`hoisted_cost` plus a fourth figure: the elements held between STAGES - every intermediate whose consumer
    depends on more sliced labels than it does is kept for all its evaluations until the consumers have run
    (`dist.StagedSlicedContraction` materialises those), so ``sum evaluations x sliced size`` over them bounds
    that executor's extra memory.

    ``outer`` = how many of the LEADING sliced labels the executor walks in a host loop (lexicographic, the first
    label outermost) instead of materialising all their values at once: what is held shrinks to one value of those
    labels at a time, and a node is re-evaluated whenever a label at or before the LAST outer label it depends on
    moves - a node below outer label 2 but not below label 1 is computed again for every value of label 1 - so the
    evaluation count of a dependency set D becomes ``prod(extent of D's inner labels) x prod(extent of outer labels
    up to the last one in D)``."""
    sl = list(slice_labels)
    sz = dict(sizes)
    for lab in sl:
        sz[lab] = 1
    cache = {}

    def evaluations(bits):
        """(evaluations over the whole contraction, evaluations alive at once)"""
        if bits not in cache:
            last_outer = -1
            for bit in range(min(outer, len(sl))):
                if bits >> bit & 1:
                    last_outer = bit
            total_, alive = 1, 1
            for bit, lab in enumerate(sl):
                if bit < outer:
                    if bit <= last_outer:
                        total_ *= sizes[lab]
                elif bits >> bit & 1:
                    total_ *= sizes[lab]
                    alive *= sizes[lab]
            cache[bits] = (total_, alive)
        return cache[bits]

    live = [(set(t), sum(1 << b for b, lab in enumerate(sl) if lab in t), True) for t in term_sets]
    total, biggest, one, held = 0, 0, 0, 0
    for step in path:
        step = tuple(sorted(step))
        if len(step) == 1:
            continue
        i, j = step
        (a, da, leaf_a), (b, db, leaf_b) = live[i], live[j]
        rest = [x for k, x in enumerate(live) if k not in (i, j)]
        new, _ = _pair_result(a, b, [x[0] for x in rest], out)
        c = _size(a | b, sz)
        total += c * max(evaluations(da | db)[0], parallel)
        one += c
        biggest = max(biggest, _size(new, sz))
        for lab_set, d, leaf in ((a, da, leaf_a), (b, db, leaf_b)):
            if not leaf and d != (da | db):
                held += evaluations(d)[1] * _size(lab_set, sz)
        live = rest + [(new, da | db, False)]
    return total, biggest, one, held


def path_time_model(term_sets, out, sizes, path, elem_bytes=4):
    """Rough device time of a path in microseconds, for choosing between candidate paths of similar multiply-add
    counts: per step the larger of (tile-padded multiply-adds at 60 T/s: rows to 128, columns to 64) and (operand
    and result bytes at 3 TB/s), plus 6 us of launch.  Constants are MI355X fp32 orders of magnitude."""
    live = [set(t) for t in term_sets]
    out_set = set(out)
    total = 0.0
    for step in path:
        step = tuple(sorted(step))
        if len(step) == 1:
            continue
        i, j = step
        a, b = live[i], live[j]
        rest = [x for k, x in enumerate(live) if k not in (i, j)]
        new, _ = _pair_result(a, b, rest, out)
        batch = _size(a & b & new, sizes)
        m = _size((a - b) & new, sizes)
        n = _size((b - a) & new, sizes)
        k = _size((a | b) - new, sizes)
        if m < n:
            m, n = n, m
        pm = -(-m // 128) * 128 if m > 32 else m
        pn = -(-n // 64) * 64 if n > 8 else n
        mac_us = batch * pm * pn * k / 60e6
        mem_us = elem_bytes * (_size(a, sizes) + _size(b, sizes) + _size(new, sizes)) / 3e6
        total += max(mac_us, mem_us) + 6.0
        live = rest + [new]
    return total


def path_profile(term_sets, out, sizes, path):
    """(flop proxy, list of every intermediate's element count) of a path."""
    live = [set(t) for t in term_sets]
    flops, inter = 0, []
    for step in path:
        step = tuple(sorted(step))
        if len(step) == 1:
            continue
        i, j = step
        rest = [x for k, x in enumerate(live) if k not in (i, j)]
        new, _ = _pair_result(live[i], live[j], rest, out)
        flops += _size(live[i] | live[j], sizes)
        inter.append(_size(new, sizes))
        live = rest + [new]
    return flops, inter


def _ssa_pairs_to_linear(pairs, n):
    """Pairs over SSA ids (inputs 0..n-1, k-th pair defines n+k) -> shrinking-list positions."""
    live = list(range(n))
    path = []
    for num, (x, y) in enumerate(pairs):
        i, j = sorted((live.index(x), live.index(y)))
        path.append((i, j))
        live.pop(j)
        live.pop(i)
        live.append(n + num)
    return path


def _dp(term_sets, out, sizes, memory_limit=None, dep=None, dep_mult=None):
    """Exact minimum-flop pairwise order by dynamic programming over operand subsets (the published
    algorithm behind opt_einsum's ``'dp'``: best tree of every subset from the best trees of its
    two-way splits).  Subsets are bit masks; the labels a subset keeps are those also needed outside
    it (other operands or the output), which makes hyperedges (labels shared by more than two
    operands) come out right.  Every split is considered, outer products included (they are
    sometimes the cheapest way to absorb small vectors), so the result equals the exhaustive search's.
    O(3^n): used up to 12 operands.

    ``dep`` / ``dep_mult`` (index slicing with slice-independent work done once, `hoisted_cost`): ``dep[k]`` is a
    bit mask of the sliced labels operand ``k`` depends on and ``dep_mult(bits)`` how many times a node depending on
    ``bits`` is evaluated; a node then costs its (sliced) index space times that number."""
    n = len(term_sets)
    out_set = set(out)
    full = (1 << n) - 1
    labels_of = {}

    def union_labels(mask):
        if mask not in labels_of:
            acc = set()
            m, k = mask, 0
            while m:
                if m & 1:
                    acc |= term_sets[k]
                m >>= 1
                k += 1
            labels_of[mask] = acc
        return labels_of[mask]

    def kept(mask):
        inside = union_labels(mask)
        outside = union_labels(full ^ mask) | out_set
        return inside & outside if mask != full else inside & out_set

    best = {1 << k: (0, None, None) for k in range(n)}   # mask -> (cost, left, right)
    # an input keeps ALL its labels until its first contraction (a label nobody else needs is summed there
    # and still spans that step's iteration space)
    keep = {1 << k: set(term_sets[k]) for k in range(n)}
    depm = {1 << k: (dep[k] if dep is not None else 0) for k in range(n)}
    for mask in range(1, full + 1):
        if mask & (mask - 1) == 0:
            continue
        lowest = mask & -mask
        sub = (mask - 1) & mask
        cand = None
        over = memory_limit is not None and mask != full and _size(kept(mask), sizes) > memory_limit
        depm[mask] = depm[lowest] | depm[mask ^ lowest]
        times = dep_mult(depm[mask]) if dep is not None else 1
        while sub:
            if sub & lowest:   # canonical: the left part holds the subset's lowest operand
                right = mask ^ sub
                cost = best[sub][0] + best[right][0] + _size(keep[sub] | keep[right], sizes) * times
                if over:
                    cost += 1 << 200    # still possible, but only if nothing else fits
                if cand is None or cost < cand[0]:
                    cand = (cost, sub, right)
            sub = (sub - 1) & mask
        if cand is not None:
            best[mask] = cand
            keep[mask] = kept(mask)
    # read the tree back as SSA pairs (children before parents)
    ids = {1 << k: k for k in range(n)}
    pairs = []

    def emit(mask):
        if mask in ids:
            return ids[mask]
        _, left, right = best[mask]
        a, b = emit(left), emit(right)
        ids[mask] = n + len(pairs)
        pairs.append((a, b))
        return ids[mask]

    emit(full)
    return _ssa_pairs_to_linear(pairs, n)


def _random_greedy(term_sets, out, sizes, repeats=32, seed=0, memory_limit=None, accept_flat=False, keep=1):
    """Greedy restarts with Boltzmann noise on the pair score (the idea of opt_einsum's
    ``'random-greedy'``): the first trial is the plain greedy path, the best path by flop count
    (then by largest intermediate) wins.  Deterministic for a given seed.  ``keep`` > 1 returns the
    ``keep`` best distinct trials, best first (starting points for `_reconfigure`)."""
    import math
    import random

    rng = random.Random(seed)
    best_path = _greedy(term_sets, out, sizes)
    best_key = path_cost(term_sets, out, sizes, best_path)
    if memory_limit is not None and best_key[1] > memory_limit:
        best_key = (best_key[0] + (1 << 200), best_key[1])
    # accept_flat ("auto"): a greedy path none of whose intermediates outgrows the largest operand (chains,
    # hubs, MPS overlaps) leaves the restarts nothing to find - skip them (a 101-leg hub: 1 ms instead of 0.8 s)
    if accept_flat and best_key[1] <= max((_size(t, sizes) for t in term_sets), default=1) and best_key[0] < (1 << 200):
        return best_path if keep == 1 else [best_path]
    trials = [(best_key, best_path)]
    out_set = set(out)
    for trial in range(1, repeats):
        temperature = 0.3 * (1 + trial % 4)
        live = [set(t) for t in term_sets]
        path = []
        while len(live) > 1:
            refs, owners = {}, {}
            for pos, t in enumerate(live):
                for s_ in t:
                    refs[s_] = refs.get(s_, 0) + 1
                    owners.setdefault(s_, []).append(pos)
            pairs = set()
            for lab, who in owners.items():      # pairs that share a label which can be summed (not an output / batch
                if len(who) > 1 and lab not in out_set:   # label: on a batch hyperedge that alone is O(n^2) pairs)
                    pairs.update(itertools.combinations(who, 2))
            if not pairs:
                for who in owners.values():
                    if len(who) > 1:
                        pairs.update(itertools.combinations(who, 2))
            if not pairs:
                order = sorted(range(len(live)), key=lambda p_: (_size(live[p_], sizes), p_))
                pairs = {tuple(sorted(order[:2]))}
            scored = []
            for i, j in sorted(pairs):
                a, b = live[i], live[j]
                new = {s_ for s_ in a | b if s_ in out_set or refs[s_] - (s_ in a) - (s_ in b) > 0}
                score = _size(new, sizes) - _size(a, sizes) - _size(b, sizes)
                # noise in the log domain, so that it matters at every scale
                noisy = math.copysign(math.log1p(abs(score)), score) - temperature * math.log(1e-12 + rng.random())
                scored.append((noisy, i, j, new))
            _, i, j, new = min(scored)
            path.append((i, j))
            live = [x for k, x in enumerate(live) if k not in (i, j)] + [new]
        key = path_cost(term_sets, out, sizes, path)
        if memory_limit is not None and key[1] > memory_limit:
            key = (key[0] + (1 << 200), key[1])
        trials.append((key, path))
        if key < best_key:
            best_key, best_path = key, path
    if keep == 1:
        return best_path
    ranked, seen = [], set()
    for key, path in sorted(trials, key=lambda kp: kp[0]):
        if key not in seen:
            seen.add(key)
            ranked.append(path)
    return ranked[:keep]


def _cluster_greedy(term_sets, out, sizes, seed=0, bound=None):
    """Grow ONE cluster from operand ``seed``: repeatedly absorb the connected operand whose absorption is
    cheapest (joint index space, then result size).  On chains and lattices this is a boundary sweep - the MPS
    zipper, the PEPS row sweep, the left-to-right walk of a chain that hangs on a batch hyperedge - which the
    pairwise greedy scores miss when merging two small neighbours looks cheaper locally."""
    n = len(term_sets)
    out_set = set(out)
    refs = {}
    for t in term_sets:
        for lab in t:
            refs[lab] = refs.get(lab, 0) + 1
    remaining = set(range(n)) - {seed}
    cluster, cid = set(term_sets[seed]), seed
    pairs, next_id = [], n
    inside = {lab: 1 for lab in term_sets[seed]}      # how many absorbed operands carry each label
    spent = 0
    while remaining:
        # connected through a label that can be summed first; an output (batch) label alone connects everything
        cands = ([t for t in remaining if (term_sets[t] & cluster) - out_set] or
                 [t for t in remaining if term_sets[t] & cluster] or list(remaining))
        best = None
        for t in cands:
            joint = cluster | term_sets[t]
            new = {lab for lab in joint if lab in out_set or refs[lab] - inside.get(lab, 0) - (lab in term_sets[t]) > 0}
            key = (_size(joint, sizes), _size(new, sizes), t)
            if best is None or key < best[0]:
                best = (key, t, new)
        key, t, new = best
        spent += key[0]
        if bound is not None and spent > bound:
            return None
        for lab in term_sets[t]:
            inside[lab] = inside.get(lab, 0) + 1
        pairs.append((cid, t))
        cid, next_id = next_id, next_id + 1
        cluster = new
        remaining.discard(t)
    return _ssa_pairs_to_linear(pairs, n)


def _best_cluster_sweep(term_sets, out, sizes, bound=None):
    """`_cluster_greedy` from a handful of seeds (both ends, the middle, the smallest and the largest operand);
    the cheapest sweep, or None when none stays under ``bound``."""
    n = len(term_sets)
    by_size = sorted(range(n), key=lambda i: (_size(term_sets[i], sizes), i))
    seeds = []
    for sd in (0, n - 1, n // 2, by_size[0], by_size[-1], n // 4, (3 * n) // 4):
        if sd not in seeds:
            seeds.append(sd)
    best = None
    for sd in seeds:
        p = _cluster_greedy(term_sets, out, sizes, sd, bound=bound if best is None else min(bound or best[0][0], best[0][0]))
        if p is None:
            continue
        key = path_cost(term_sets, out, sizes, p)
        if best is None or key < best[0]:
            best = (key, p)
    return None if best is None else best[1]


def _reconfigure(term_sets, out, sizes, path, max_leaves=8, rounds=8, memory_limit=None, slice_labels=None,
                 parallel=1):
    """Subtree reconfiguration of a pairwise contraction tree (the refinement step of hyper-optimised
    path finders): for every internal node, cut out the subtree spanned by up to ``max_leaves`` of its
    descendants (largest intermediates expanded first), solve that small network EXACTLY with the
    subset DP - its operands are the cut-off subtrees, its output the node's own index set - and keep
    the result when it is cheaper (and, under a ``memory_limit``, does not push an intermediate of the
    subtree past both the limit and what the subtree already had).  Never worse than the input path in
    flops; returns a linear path.  8 x 8 PEPS, D = 8: 1.2e12 -> 2.2e11 multiply-adds in under a second
    (the hand-written row sweep costs 3.2e11).

    ``slice_labels`` (with ``sizes`` the UNSLICED extents): optimise the work of the index-sliced contraction with
    slice-independent work done once (`hoisted_cost`) - a node costs its sliced index space times the number of
    joint values of the sliced labels its subtree depends on (at least ``parallel``, see `hoisted_cost`)."""
    n = len(term_sets)
    if n < 4:
        return list(path)
    sl = list(slice_labels or ())
    full_sizes = sizes
    if sl:
        sizes = dict(sizes)
        for lab in sl:
            sizes[lab] = 1
    mult_cache = {}

    def dep_mult(bits):
        if bits not in mult_cache:
            m = 1
            for b, lab in enumerate(sl):
                if bits >> b & 1:
                    m *= full_sizes[lab]
            mult_cache[bits] = max(m, parallel)
        return mult_cache[bits]

    out_set = set(out)
    total = {}
    for t in term_sets:
        for lab in set(t):
            total[lab] = total.get(lab, 0) + 1

    # tree as dicts keyed by node id: leaves 0..n-1, internal nodes get fresh ids
    kids, count = {}, {i: {lab: 1 for lab in set(t)} for i, t in enumerate(term_sets)}
    depb = {i: sum(1 << b for b, lab in enumerate(sl) if lab in t) for i, t in enumerate(term_sets)}
    live, nxt = list(range(n)), n
    for step in path:
        step = tuple(sorted(step))
        if len(step) == 1:
            continue
        a, b = live[step[0]], live[step[1]]
        live = [x for k, x in enumerate(live) if k not in step] + [nxt]
        kids[nxt] = (a, b)
        c = dict(count[a])
        for lab, v in count[b].items():
            c[lab] = c.get(lab, 0) + v
        count[nxt] = c
        depb[nxt] = depb[a] | depb[b]
        nxt += 1
    root = live[0]

    def labels(v):   # index set of node v: labels of its leaves that are still needed outside the subtree
        if v < n:      # an operand keeps all its labels (one that nobody else has is summed when it is first used)
            return set(term_sets[v])
        return {lab for lab, c in count[v].items() if lab in out_set or c < total[lab]}

    lab_cache = {}

    def L(v):
        if v not in lab_cache:
            lab_cache[v] = labels(v)
        return lab_cache[v]

    def node_cost(v):
        a, b = kids[v]
        return _size(L(a) | L(b), sizes) * (dep_mult(depb[v]) if sl else 1)

    for _ in range(rounds):
        improved = False
        # top-down order: parents first, so a rewritten subtree is revisited through its new nodes next round
        order, stack = [], [root]
        while stack:
            v = stack.pop()
            if v in kids:
                order.append(v)
                stack.extend(kids[v])
        for v in order:
            if v not in kids:
                continue
            frontier, inner = [v], []
            while len(frontier) < max_leaves:
                cand = [x for x in frontier if x in kids]
                if not cand:
                    break
                x = max(cand, key=lambda y: (_size(L(y), sizes), y))
                frontier.remove(x)
                frontier.extend(kids[x])
                inner.append(x)
            if len(frontier) < 3:
                continue
            old_cost = sum(node_cost(x) for x in inner)
            sub_sets = [L(x) for x in frontier]
            if sl:
                sub_dep = [depb[x] for x in frontier]
                sub_path = _dp(sub_sets, L(v), sizes, memory_limit=memory_limit, dep=sub_dep, dep_mult=dep_mult)
                _plain, new_big = path_cost(sub_sets, L(v), sizes, sub_path)
                new_cost, cur_d = 0, list(zip(sub_sets, sub_dep))
                for st in sub_path:          # the same objective, read off the path
                    i_, j_ = sorted(st)
                    (a_, da_), (b_, db_) = cur_d[i_], cur_d[j_]
                    rest_ = [x for q, x in enumerate(cur_d) if q not in (i_, j_)]
                    kept_, _ = _pair_result(a_, b_, [x[0] for x in rest_], L(v))
                    new_cost += _size(a_ | b_, sizes) * dep_mult(da_ | db_)
                    cur_d = rest_ + [(kept_, da_ | db_)]
            else:
                sub_path = _dp(sub_sets, L(v), sizes, memory_limit=memory_limit)
                new_cost, new_big = path_cost(sub_sets, L(v), sizes, sub_path)
            if new_cost >= old_cost:
                continue
            if memory_limit is not None and new_big > max(memory_limit, max(_size(L(x), sizes) for x in inner)):
                continue
            improved = True
            for x in inner:          # drop the old interior (v keeps its id and index set)
                del kids[x]
                if x != v:
                    del count[x]
                    lab_cache.pop(x, None)
            cur = list(frontier)
            for k, step in enumerate(sub_path):
                i, j = sorted(step)
                a, b = cur[i], cur[j]
                last = k == len(sub_path) - 1
                nid = v if last else nxt
                if not last:
                    nxt += 1
                    c = dict(count[a])
                    for lab, val in count[b].items():
                        c[lab] = c.get(lab, 0) + val
                    count[nid] = c
                    depb[nid] = depb[a] | depb[b]
                kids[nid] = (a, b)
                cur = [x for q, x in enumerate(cur) if q not in (i, j)] + [nid]
        if not improved:
            break

    # back to a linear path: post-order over the tree
    ssa, done = [], {}
    stack = [(root, False)]
    next_id = n
    while stack:
        v, seen = stack.pop()
        if v not in kids:
            done[v] = v
            continue
        if not seen:
            stack.append((v, True))
            stack.extend((c, False) for c in kids[v])
        else:
            a, b = kids[v]
            ssa.append((done[a], done[b]))
            done[v] = next_id
            next_id += 1
    return _ssa_pairs_to_linear(ssa, n)


def find_path(terms, out, sizes, optimize, memory_limit=None):
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
        # opt_einsum: no optimisation = ONE step over every operand (a single einsum); the engine orders it
        # pairwise when the contraction list is lowered (einsum.lower_contraction_list)
        return [tuple(range(n))]
    if name in ("auto", "auto-hq"):
        # few operands: exact; up to 12: exact by subset DP; beyond: see the last branch
        if n < 5:
            name = "optimal"
        elif n <= 12:
            name = "dp"
        else:
            # greedy with noisy restarts (8 up to 64 operands, 4 up to 256, the single greedy pass beyond, as
            # opt_einsum; "auto-hq": 128 at any size), then subtree reconfiguration of the winner - unless the
            # greedy path is already flat (no intermediate outgrows the largest operand: chains, hubs, MPS
            # overlaps), which is taken as it is.  All of it is cached per network by the caller.
            hq = name == "auto-hq"
            repeats = 128 if hq else (8 if n <= 64 else (4 if n <= 256 else 1))
            starts = _random_greedy(sets, out, sizes, repeats=repeats, memory_limit=memory_limit, accept_flat=not hq,
                                    keep=4 if hq else 2)[:4 if hq else 1]
            path = starts[0]
            biggest_in = max((_size(t, sizes) for t in sets), default=1)
            if not hq and path_cost(sets, out, sizes, path)[1] <= biggest_in:
                return path
            # a second kind of start: the best single-cluster sweep (boundary sweeps of chains and lattices); anything
            # costing more than 100 x the pairwise result is a blown-up seed and is dropped on the way
            sweep = _best_cluster_sweep(sets, out, sizes, bound=100 * path_cost(sets, out, sizes, path)[0])
            if sweep is not None:
                starts = list(starts) + [sweep]
            # refining a start that is several times dearer than the best one rarely catches up: keep those within 3 x
            costs = [path_cost(sets, out, sizes, st)[0] for st in starts]
            starts = [st for st, c in zip(starts, costs) if c <= 3 * min(costs)]
            # the refinement is a local search on a rugged landscape: the best start is not always the best finish,
            # so every start that is left is refined and the result with the least MODELLED DEVICE TIME is kept
            # (`path_time_model`: tile padding, bytes and launches, not multiply-adds alone)
            better = min((_reconfigure(sets, out, sizes, st, max_leaves=10 if hq else (8 if n <= 256 else 6),
                                       rounds=12 if hq else 8, memory_limit=memory_limit) for st in starts),
                         key=lambda q: (path_time_model(sets, out, sizes, q), path_cost(sets, out, sizes, q)))
            # fewer flops must not buy an intermediate the engine cannot hold (2^31 elements) when the start fits
            if path_cost(sets, out, sizes, better)[1] >= (1 << 31) > path_cost(sets, out, sizes, path)[1]:
                return path
            return better
    if name == "optimal" and n <= 8 and memory_limit is None:
        return _optimal(sets, out, sizes)
    if name in ("optimal", "dp", "branch-all", "branch-2", "branch-1"):
        if n <= 12:
            return _dp(sets, out, sizes, memory_limit=memory_limit)
        return _random_greedy(sets, out, sizes, repeats=64, memory_limit=memory_limit)
    if name.startswith("random-greedy"):
        reps = 32
        if "-" in name[len("random-greedy"):]:
            reps = max(1, int(name.rsplit("-", 1)[1]))
        return _random_greedy(sets, out, sizes, repeats=reps, memory_limit=memory_limit)
    if name in ("greedy", "eager", "opportunistic"):
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
    limit = None
    if memory_limit not in (None, -1):
        # opt_einsum semantics: a number of elements, or 'max_input' = the largest operand
        limit = max(_size(set(t), sizes) for t in terms) if memory_limit == "max_input" else int(memory_limit)
    path = find_path(terms, out, sizes, optimize, memory_limit=limit)
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
