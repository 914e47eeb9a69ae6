#!/bin/bash
# Development: the 8 x 8 D = 16 PEPS contraction under experiment builds (make EXP=n).
set -o pipefail
for e in $1; do
  lib=contractn_amd/lib/libctn_hip_exp$e.so; [ $e = 0 ] && lib=contractn_amd/lib/libctn_hip.so
  CTN_LIB_PATH=$PWD/$lib timeout -k 10 300 python bench.py --config peps --bond 16 --max-intermediate 268435456 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/p16exp_$e.json 2> gpurun_out/p16exp_$e.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/p16exp_$e.json").read().strip().splitlines()[-1])
print("exp $e", d["ms_per_step"], d["roofline"]["kernel"][:40], d["roofline"]["frac"])
PY
done
