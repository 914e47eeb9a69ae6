#!/bin/bash
# Development: the 8 x 8 D = 8 PEPS contraction under experiment builds (make EXP=n).
set -o pipefail
for e in $1; do
  lib=contractn_amd/lib/libctn_hip_exp$e.so; [ $e = 0 ] && lib=contractn_amd/lib/libctn_hip.so
  CTN_LIB_PATH=$PWD/$lib timeout -k 10 200 python bench.py --config peps --bond 8 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/pexp_$e.json 2> gpurun_out/pexp_$e.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/pexp_$e.json").read().strip().splitlines()[-1])
print("exp $e", d["ms_per_step"], d["roofline"]["kernel"][:40], d["roofline"]["frac"])
PY
done
