#!/bin/bash
# dev: gpurun client with retries while the pod's GPU slots are busy (exit 3 = nothing charged).  usage: tools/gpu.sh <timeout-s> '<command>'
t=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 60
done
exit 3
