#!/usr/bin/env python3
"""Regenerate the sliced plans that ship in contractn_amd/plans/ (host only, minutes): the staged plans of the two
8 x 8 PEPS benchmarks (D = 8 with >= 64 slices; D = 16 under a 2^28- and a 2^32-element cap) and of the 5 x 6, D = 16 test
network, each the best of many seeds searched in parallel processes, stored under the cache key of the DEFAULT call
(`dist.staged_plan(einstr, shapes, min_slices=..., max_intermediate=...)`), so that a GPU box never searches.

    python tools/make_plans.py [--seeds 64] [--procs 8]
"""
import argparse
import hashlib
import json
import os
import sys
from multiprocessing import Pool

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

JOBS = [  # (rows, cols, bond, min_slices, max_intermediate)
    (8, 8, 8, 64, None),
    (8, 8, 16, 64, 2 ** 28),
    (8, 8, 16, 64, 2 ** 32),
    (5, 6, 16, 16, None),
]


def one(job):
    import bench
    from contractn_amd import dist

    (rows, cols, bond, ms, mi), seed = job
    einstr, shapes, _ops = bench.peps_network(rows, cols, bond)
    try:
        labels, path, rep = dist.choose_staged_slices(einstr, shapes, min_slices=ms, max_intermediate=mi, seeds=1, first_seed=seed)
    except ValueError:
        return None
    return rep["modelled_overhead_at_parallel"], seed, labels, path, rep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=64)
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--only", type=int, nargs="*", default=None, help="indices into JOBS")
    ap.add_argument("--first-seed", type=int, default=0, help="search seeds first .. first + seeds - 1; a shipped plan is only replaced by a better one")
    args = ap.parse_args()
    import bench

    for k, job in enumerate(JOBS):
        if args.only is not None and k not in args.only:
            continue
        rows, cols, bond, ms, mi = job
        with Pool(args.procs) as pool:
            res = [r for r in pool.map(one, [(job, s) for s in range(args.first_seed, args.first_seed + args.seeds)], chunksize=1) if r is not None]
        res.sort(key=lambda r: (r[0], r[1]))
        key, seed, labels, path, rep = res[0]
        einstr, shapes, _ops = bench.peps_network(rows, cols, bond)
        shapes = [tuple(int(d) for d in sh) for sh in shapes]
        h = hashlib.sha1(json.dumps(["staged", einstr, shapes, int(ms), mi, []], ensure_ascii=True).encode()).hexdigest()[:16]
        fname = os.path.join(ROOT, "contractn_amd", "plans", f"staged_{h}.json")
        rep = dict(rep, search=f"best of {len(res)} seeds (tools/make_plans.py), seed {seed}")
        if args.first_seed and os.path.exists(fname):
            old = json.load(open(fname))["report"]
            if old.get("modelled_overhead_at_parallel", 1e9) <= key:
                print(f"{rows}x{cols} D={bond}: shipped plan (x{old['modelled_overhead_at_parallel']:.3f} at {old['parallel']} ranks) stays; "
                      f"best of seeds {args.first_seed}..{args.first_seed + args.seeds - 1}: x{key:.3f}", flush=True)
                continue
        with open(fname, "w") as fh:
            json.dump({"einsum_str": einstr, "shapes": shapes, "min_slices": int(ms), "max_intermediate": mi,
                       "labels": list(labels), "path": [list(p) for p in path], "report": rep}, fh)
        print(f"{rows}x{cols} D={bond}: {fname}\n   labels {labels} slices {rep['slices']} work x{rep['work_overhead']:.3f} "
              f"(plain x{rep['plain_overhead']:.3f}; at {rep['parallel']} ranks x{rep['modelled_overhead_at_parallel']:.3f}) "
              f"peak {rep['largest_intermediate']} held {rep['held_between_stages']}; seeds: "
              + " ".join(f"{r[0]:.2f}" for r in res[:12]), flush=True)


if __name__ == "__main__":
    main()
