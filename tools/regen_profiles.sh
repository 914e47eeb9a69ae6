#!/bin/bash
# Regenerate the evidence under profiles/ on a GPU box (run from the repo root):
#   gpurun --timeout 1200 -- 'bash tools/regen_profiles.sh'      then, back in the container,
#   python profiles/make_summary.py r01 gpurun_out/prof_r01 gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq
#   cp gpurun_out/bench_default.json profiles/r01_bench.json; cp gpurun_out/bench_f64.json profiles/r01_bench_f64.json
# Counters are collected in their own passes (no trace domains next to --pmc).
set -e -o pipefail
mkdir -p gpurun_out
B="--no-cpu-baseline"
timeout -k 10 400 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
timeout -k 10 300 python bench.py --dtype f64 --replicas 256 --cpu-seconds 5 > gpurun_out/bench_f64.json 2> gpurun_out/bench_f64.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r01 -- python3 bench.py $B > gpurun_out/prof_r01.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 $B > gpurun_out/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 2 --warmup 1 $B > gpurun_out/pmc_write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_sq -- python3 bench.py --steps 2 --warmup 1 $B > gpurun_out/pmc_sq.log 2>&1
tail -1 gpurun_out/bench_default.json | cut -c1-400
