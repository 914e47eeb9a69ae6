#!/bin/bash
# Regenerate the evidence under profiles/ on a GPU box (run from the repo root):
#   gpurun --timeout 1200 -- 'bash tools/regen_profiles.sh r03'      then, back in the container,
#   python profiles/make_summary.py r03 gpurun_out/prof_r03 gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq gpurun_out/bench_default.json
#   cp gpurun_out/bench_default.json profiles/r03_bench.json; cp gpurun_out/bench_f64.json profiles/r03_bench_f64.json
# Counters are collected in their own passes (no trace domains next to --pmc); the program itself follows `--`.
set -o pipefail
tag=${1:-r03}
mkdir -p gpurun_out
B="--no-cpu-baseline --no-peps"
R=$GRAFT_REPO_ROOT
run() { name=$1; secs=$2; shift 2; echo "== $name"; timeout -k 10 $secs "$@" > $R/gpurun_out/$name.log 2>&1; rc=$?; echo "== $name rc=$rc"; [ $rc -ge 124 ] && exit $rc; return 0; }
timeout -k 10 500 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench_default rc=$?"
timeout -k 10 300 python bench.py --dtype f64 --replicas 256 --cpu-seconds 5 --no-peps > gpurun_out/bench_f64.json 2> gpurun_out/bench_f64.err; echo "bench_f64 rc=$?"
cd /tmp && export TMPDIR=/tmp
run prof_$tag 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/bench.py $B
run pmc_fetch 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 $B
run pmc_write 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 $B
run pmc_sq 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_sq -- python3 $R/bench.py --steps 2 --warmup 1 $B
cd $R
# the secondary configs and the sliced 8 x 8 PEPS (copied to profiles/<tag>_config_bench.jsonl, <tag>_peps_D8_bench.json /
# _steps.json / _kernel_stats.csv, <tag>_peps_D16_bench.json / _steps.json / _kernel_stats.csv afterwards)
timeout -k 10 500 python tools/config_bench.py > gpurun_out/${tag}_config_bench.jsonl 2> gpurun_out/config_bench.err; echo "config_bench rc=$?"
timeout -k 10 300 python bench.py --config peps --bond 8 --steps 20 --warmup 3 --dump-steps gpurun_out/${tag}_peps_D8_steps.json > gpurun_out/${tag}_peps_D8_bench.json 2> gpurun_out/peps_D8.err; echo "peps D8 rc=$?"
timeout -k 10 500 python bench.py --config peps --bond 16 --max-intermediate 4294967296 --steps 1 --warmup 1 --cross-check --no-cpu-baseline --dump-steps gpurun_out/${tag}_peps_D16_steps.json > gpurun_out/${tag}_peps_D16_bench.json 2> gpurun_out/peps_D16.err; echo "peps D16 rc=$?"
cd /tmp
run prof_peps8 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_peps8 -- python3 $R/bench.py --config peps --bond 8 --steps 20 --warmup 3 --no-cpu-baseline
run prof_peps16 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_peps16 -- python3 $R/bench.py --config peps --bond 16 --max-intermediate 4294967296 --steps 1 --warmup 1 --no-cpu-baseline --event-passes 0
run pmc_sq_peps16 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_sq_peps16 -- python3 $R/bench.py --config peps --bond 16 --max-intermediate 4294967296 --steps 1 --warmup 1 --no-cpu-baseline --event-passes 0
cd $R
tail -1 gpurun_out/bench_default.json | cut -c1-300
