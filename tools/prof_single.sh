#!/bin/bash
# rocprofv3 kernel trace of ONE headline network in flight (bench.py --replicas 1): per-kernel durations without launch gaps
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$1 -- python3 $GRAFT_REPO_ROOT/bench.py --replicas 1 --steps 20 --warmup 3 --no-cpu-baseline --no-peps --event-passes 0 > $GRAFT_REPO_ROOT/gpurun_out/$1.log 2>&1
f=$(ls $GRAFT_REPO_ROOT/gpurun_out/$1/*/*kernel_stats.csv | tail -1)
head -8 $f | cut -c1-160
