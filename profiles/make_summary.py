#!/usr/bin/env python3
"""Turn raw rocprofv3 outputs (gpurun_out/, scratch) into the committed per-round summaries.

    python profiles/make_summary.py r02 gpurun_out/prof_r02 gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq \
        gpurun_out/bench_default.json

The last argument is the bench line of the same command in the same gpurun call: its workload name and dominant-kernel
label are stored next to the counters, and bench.py attaches `roofline.traffic` only to a run that matches both.

Writes profiles/<round>_kernel_stats.csv (rocprofv3 --kernel-trace --stats, ctn kernels only),
profiles/<round>_pmc_summary.json (per-kernel counter means) and profiles/pmc_traffic.json
(HBM bytes per launch of the dominant kernel, FETCH_SIZE doubled per MI355X_MICROARCH.md sec. HBM:
on gfx950 FETCH_SIZE tallies 128-B requests at 64 B; both counters are in KiB).
"""
import collections
import csv
import glob
import json
import os
import sys


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "*", "*" + suffix))
    return max(hits, key=os.path.getmtime) if hits else None  # scratch dirs keep earlier runs: newest wins


def counters(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(path)):
        agg[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"launches": len(next(iter(cs.values())))}
            for k, cs in agg.items()}


def main():
    tag, stats_dir, fetch_dir, write_dir, sq_dir = sys.argv[1:6]
    bench = {}
    if len(sys.argv) > 6:
        bench = json.loads(open(sys.argv[6]).read().strip().splitlines()[-1])
    import subprocess

    try:
        sha = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                             cwd=os.path.dirname(os.path.abspath(__file__))).stdout.strip() or "unknown"
    except OSError:
        sha = "unknown"
    here = os.path.dirname(os.path.abspath(__file__))
    rows = list(csv.reader(open(find(stats_dir, "kernel_stats.csv"))))
    with open(os.path.join(here, f"{tag}_kernel_stats.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(rows[0])
        for r in rows[1:]:
            if "ctn::" in r[0]:
                w.writerow(r)
    summary = {}
    for name, d in (("fetch", fetch_dir), ("write", write_dir), ("sq", sq_dir)):
        for kern, vals in counters(find(d, "counter_collection.csv")).items():
            if "ctn::" in kern:
                summary.setdefault(kern, {}).update(vals)
    for kern, v in summary.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            v["hbm_bytes_per_launch"] = (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v and v.get("GRBM_GUI_ACTIVE"):
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs on the chip
            v["mfma_busy_frac"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
    json.dump(summary, open(os.path.join(here, f"{tag}_pmc_summary.json"), "w"), indent=1, sort_keys=True)
    stat_rows = {r[0]: float(r[2]) for r in rows[1:] if "ctn::" in r[0]}
    dom = max(stat_rows, key=stat_rows.get)
    # the headline's dominant kernel as the bench line names it (the secondary configs of the same run launch long
    # GEMMs of their own: by total time alone one of those can come first)
    label = (bench.get("roofline", {}).get("kernel") or "").split(" ")[0].split("<")[0]
    named = [k for k in stat_rows if label and ("ctn::" + label + "(") in k.replace("<", "(")]
    if named:
        dom = max(named, key=stat_rows.get)
    traffic = {"kernel": dom, "round": tag, "git_sha": sha,
               "workload": bench.get("config", {}).get("workload"),
               "kernel_label": bench.get("roofline", {}).get("kernel"),
               "algorithmic_bytes_per_launch": bench.get("roofline", {}).get("algorithmic_bytes_per_launch"),
               "hbm_bytes_per_launch": summary.get(dom, {}).get("hbm_bytes_per_launch"),
               "note": "(2*FETCH_SIZE + WRITE_SIZE) KiB, separate --pmc passes, mean over launches"}
    json.dump(traffic, open(os.path.join(here, "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(traffic))
    for r in rows[1:]:
        if "ctn::" in r[0]:
            print(r[0][:60], "calls", r[1], "avg_ns", r[3], "pct", r[4])


if __name__ == "__main__":
    main()
