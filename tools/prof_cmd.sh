#!/bin/bash
# rocprofv3 kernel trace of an arbitrary python tool:  bash tools/prof_cmd.sh <name> <script.py> [args...]
# (the program itself follows `--`: python3 <absolute script path>, nothing that re-execs)
name=$1; shift
script=$GRAFT_REPO_ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$name -- python3 $script "$@" > $GRAFT_REPO_ROOT/gpurun_out/$name.log 2>&1
f=$(ls $GRAFT_REPO_ROOT/gpurun_out/$name/*/*kernel_stats.csv | tail -1)
head -12 $f | cut -c1-170
