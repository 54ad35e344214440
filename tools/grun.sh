#!/usr/bin/env bash
# gpurun with patience: retries while the pool reports "no box / slot free" (nothing is charged for those attempts).
# usage: tools/grun.sh <timeout-seconds> '<command>'
t=$1; shift
for i in $(seq 1 40); do
  out=$(/usr/local/graft/bin/gpurun --timeout "$t" -- "$@" 2>&1)
  if echo "$out" | grep -q "status=transient"; then sleep 60; continue; fi
  echo "$out"; exit 0
done
echo "$out"; exit 3
