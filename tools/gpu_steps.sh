#!/bin/bash
# Run a list of GPU steps one after another on a gpurun box:  bash tools/gpu_steps.sh "<name>|<seconds>|<command>" ...
# Each step runs under `timeout -k 10 <seconds>`, stdout/stderr go to gpurun_out/<name>.log.  A step that FAILS
# is recorded and the next one still runs; a step that is KILLED at its limit (or dies on a signal) ends the
# call - nothing more is started on a GPU that may be wedged.
mkdir -p gpurun_out
for spec in "$@"; do
  name="${spec%%|*}"; rest="${spec#*|}"; secs="${rest%%|*}"; cmd="${rest#*|}"
  echo "== $name (limit ${secs}s): $cmd"
  timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "== $name rc=$rc"
  tail -n 3 "gpurun_out/$name.log" | cut -c1-600
  if [ $rc -ge 124 ]; then echo "== $name was killed (rc=$rc): stopping here"; exit $rc; fi
done
exit 0
