#!/bin/bash
# Development: BASELINE config 4 legs of bench.py (configs object) under an environment, e.g. CTN_G_BIG_MIN_K=1024
python - <<PY
import json, subprocess, sys, os
out = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--replicas", "8", "--sites", "12", "--bond", "64",
                      "--no-peps", "--no-batched", "--no-cpu-baseline", "--no-latency"], capture_output=True, text=True)
line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
for k, v in line["configs"].items():
    if k.startswith("cfg4"):
        print(os.environ.get("CTN_G_BIG_MIN_K"), k, v.get("ms_per_contraction"), v.get("frac_of_mfma_peak"), [(s["kernel"][:28], s["ms"]) for s in v.get("steps", [])], v.get("error"))
PY
