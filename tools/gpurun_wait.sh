#!/bin/bash
# gpurun with a wait for a free slot: repeats the call only while gpurun answers 3 ("no box or slot free right now — nothing charged");
# any other exit code (the command ran, was refused, timed out) ends it.  usage: bash tools/gpurun_wait.sh <timeout_s> '<command>'
t=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@"; rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
