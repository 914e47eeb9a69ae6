#!/bin/bash
# Development: parity + quick headline bench of k_zip_f32 experiment builds (make EXP=n), stamps of the STAMPS builds.
#   gpurun -- 'bash tools/zip_exp.sh "0 1 2 3" "8 9 10 11"'
set -o pipefail
mkdir -p gpurun_out
for e in $1; do
  lib=contractn_amd/lib/libctn_hip_exp$e.so; [ $e = 0 ] && lib=contractn_amd/lib/libctn_hip.so
  CTN_LIB_PATH=$PWD/$lib timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -q -k zip > gpurun_out/zexp_t$e.log 2>&1 || { echo "exp $e: tests FAILED"; tail -5 gpurun_out/zexp_t$e.log; continue; }
  CTN_LIB_PATH=$PWD/$lib timeout -k 10 200 python bench.py --no-peps --no-batched --no-cpu-baseline --no-latency --steps 6 > gpurun_out/zexp_b$e.json 2> gpurun_out/zexp_b$e.err || { echo "exp $e: bench FAILED"; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/zexp_b$e.json").read().strip().splitlines()[-1])
print("exp $e", d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"])
PY
done
for e in $2; do
  CTN_LIB_PATH=$PWD/contractn_amd/lib/libctn_hip_exp$e.so timeout -k 10 200 python tools/stamps_zip.py 512 > gpurun_out/zexp_s$e.log 2>&1; echo "stamps $e: $(tail -1 gpurun_out/zexp_s$e.log)"
done
