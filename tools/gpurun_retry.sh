#!/bin/bash
# gpurun with retries while every GPU slot of the pod is busy (exit code 3: nothing charged)
#   tools/gpurun_retry.sh <timeout-seconds> '<command>'
t=$1; shift
for k in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"; rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
