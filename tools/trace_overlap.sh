#!/usr/bin/env bash
# kernel trace of the default (two-stream) bench step -> idle gaps / per-queue spans (tools/trace_gaps.py)
set -uo pipefail
OUT=${1:-gpurun_out/overlap}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export HIP_FORCE_DEV_KERNARG=1   # (bench.py sets it itself, but under rocprofv3 the profiler initialises HIP before python starts)
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-fp32-line > $OUT/line.json 2> $OUT/err.txt || { tail -5 $OUT/err.txt; exit 1; }
MS=$(python3 -c "import json;print(json.load(open('$OUT/line.json'))['ms_per_step'])")
python3 tools/trace_gaps.py $OUT/tr/*/*kernel_trace.csv $MS > $OUT/gaps.txt
rm -rf $OUT/tr
cat $OUT/gaps.txt
