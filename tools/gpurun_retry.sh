#!/bin/bash
# Development helper: `gpurun` again while the pool answers "no box / no slot free" (exit code 3: nothing ran, nothing
# was charged).  Any other outcome - success, failure, refusal - is returned as it is: a command that RAN is never retried.
#   tools/gpurun_retry.sh <timeout-seconds> '<command>'
limit=$1; shift
for attempt in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$limit" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
