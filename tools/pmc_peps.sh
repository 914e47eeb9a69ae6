#!/bin/bash
# HBM traffic counters of the sliced 8x8 PEPS bench (separate --pmc passes; the program itself follows `--`)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_peps_$c -- python3 $R/bench.py --config peps --steps 2 --warmup 2 --no-cpu-baseline --event-passes 0 > $R/gpurun_out/pmc_peps_$c.log 2>&1
  echo "$c rc=$?"
done
