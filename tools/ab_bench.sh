#!/bin/bash
# A/B of kernel experiment builds inside ONE gpurun call (box-to-box variance is +-3 %):
#   tools/ab_bench.sh "<bench args>" exp-suffix...     ("base" = the production library)
args="$1"; shift
for pass in 1 2; do
  for v in "$@"; do
    lib=contractn_amd/lib/libctn_hip.so
    [ "$v" != base ] && lib=contractn_amd/lib/libctn_hip_$v.so
    CTN_LIB_PATH=$PWD/$lib timeout -k 10 280 python bench.py $args --no-cpu-baseline > gpurun_out/ab_$v.log 2>&1 || { echo "$v FAILED"; tail -3 gpurun_out/ab_$v.log; exit 1; }
    python - "$v" <<PY
import json, sys
d = json.loads(open("gpurun_out/ab_%s.log" % sys.argv[1]).read().strip().split("\n")[-1])
print(sys.argv[1], d["value"], d["achieved_tflops"], d["roofline"]["avg_launch_us"], flush=True)
PY
  done
done
