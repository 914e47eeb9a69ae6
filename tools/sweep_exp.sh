#!/bin/bash
# Development: the batched-MPS secondary under k_sweep_f32 experiment builds (make EXP=n).
set -o pipefail
for e in $1; do
  lib=contractn_amd/lib/libctn_hip_exp$e.so; [ $e = 0 ] && lib=contractn_amd/lib/libctn_hip.so
  CTN_LIB_PATH=$PWD/$lib timeout -k 10 200 python bench.py --no-peps --no-cpu-baseline --no-latency --steps 2 --warmup 1 --replicas 8 > gpurun_out/swexp_$e.json 2> gpurun_out/swexp_$e.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/swexp_$e.json").read().strip().splitlines()[-1])
b=d["batched_mps"]; print("exp $e", b.get("ms_per_pass"), b.get("frac_of_mfma_peak"), b.get("error"))
PY
done
