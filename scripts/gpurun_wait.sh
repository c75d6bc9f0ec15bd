#!/bin/bash
# usage: scripts/gpurun_wait.sh <timeout> '<command>'  -- gpurun, retried only while no GPU slot is free (exit code 3: nothing
# ran, nothing was charged).  Any other outcome is returned as it is.
T=$1; shift
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
