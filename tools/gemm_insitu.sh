#!/usr/bin/env bash
# in-situ per-shape GEMM table of one training step (single stream): see tools/gemm_insitu.py
set -uo pipefail
OUT=${1:-gpurun_out/insitu}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export HIP_FORCE_DEV_KERNARG=1   # (bench.py sets it itself, but under rocprofv3 the profiler initialises HIP before python starts)
mkdir -p $OUT
export MISSM_GEMM_LOG=$OUT/shapes.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-fp32-line --serial-streams > $OUT/line.json 2> $OUT/err.txt || { tail -5 $OUT/err.txt; exit 1; }
python3 tools/gemm_insitu.py $OUT/shapes.log $OUT/tr/*/*kernel_trace.csv 4 > $OUT/table.txt
python3 tools/trace_by_grid.py $OUT/tr/*/*kernel_trace.csv 4 > $OUT/by_grid.txt
cp $OUT/tr/*/*kernel_stats.csv $OUT/kernel_stats.csv
rm -rf $OUT/tr
cat $OUT/table.txt
