#!/usr/bin/env bash
# per-shape GEMM traffic inside the step (single stream): see tools/gemm_traffic_insitu.py.  Usage on the GPU box: bash tools/gemm_traffic_insitu.sh OUTDIR
set -uo pipefail
OUT=${1:-gpurun_out/gtraffic}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export HIP_FORCE_DEV_KERNARG=1
mkdir -p $OUT
P="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-fp32-line --serial-streams"
MISSM_GEMM_LOG=$OUT/shapes_f.log rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/f -- $P > /dev/null 2> $OUT/f.err || { tail -5 $OUT/f.err; exit 1; }
MISSM_GEMM_LOG=$OUT/shapes_w.log rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/w -- $P > /dev/null 2> $OUT/w.err || { tail -5 $OUT/w.err; exit 1; }
python3 tools/gemm_traffic_insitu.py $OUT/shapes_f.log $OUT/f/*/*counter_collection.csv $OUT/shapes_w.log $OUT/w/*/*counter_collection.csv 2 > $OUT/table.txt
rm -rf $OUT/f $OUT/w
cat $OUT/table.txt
