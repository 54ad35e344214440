#!/usr/bin/env bash
# gpurun with patience: retries while the pool reports "no box / slot free" (nothing is charged for those attempts).
# usage: tools/grun.sh <timeout-seconds> '<command>'
# Exit status: gpurun's own (0 ok, the command's failure code, 2 refused, ...); 3 when every retry found no box.
t=$1; shift
for i in $(seq 1 40); do
  out=$(/usr/local/graft/bin/gpurun --timeout "$t" -- "$@" 2>&1)
  rc=$?
  if [ $rc -eq 3 ] || echo "$out" | grep -q "status=transient"; then sleep 60; continue; fi
  echo "$out"; exit $rc
done
echo "$out"; exit 3
