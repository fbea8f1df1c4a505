#!/bin/bash
# usage: tools/gpu/submit.sh <name> <timeout_s> '<command>'   -> gpurun_out/<name>.out ; retries while the pod's GPU slots are busy
name=$1; to=$2; shift 2
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $to -- "$@" > /root/repo/gpurun_out/$name.out 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then echo "rc=$rc" >> /root/repo/gpurun_out/$name.out; exit $rc; fi
  sleep 45
done
