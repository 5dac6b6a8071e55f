#!/bin/bash
# Dev helper (this container): run a command on an MI355X box through gpurun, retrying while no box / slot is free (exit code 3).
# usage: tools/gpu.sh <timeout_s> '<command>'        (log: /tmp/gpu_last.log)
t=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@" > /tmp/gpu_last.log 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then cat /tmp/gpu_last.log | tail -60; exit $rc; fi
  sleep 45
done
echo "no GPU slot after 40 tries"; exit 3
