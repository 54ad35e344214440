#!/usr/bin/env bash
# The round's judged profiles, all from ONE box (run on the GPU box from the repo root):
#   kernel stats of the serial-stream bench, the FETCH_SIZE / WRITE_SIZE / MFMA-busy PMC passes (-> per-family HBM-side traffic and
#   utilisation), the per-shape GEMM time and traffic tables inside the step, SQ counters of the attention kernels, the two-stream
#   overlap summary, the default bench line (with the fp32 line and the CPU baseline) and the bench lines of configs[1] / [3] / [4].
# usage: MISSM_COMMIT=<sha> bash tools/profile_round.sh [round tag, default r03]
set -uo pipefail
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export HIP_FORCE_DEV_KERNARG=1   # (bench.py sets it itself, but under rocprofv3 the profiler initialises HIP before python starts)
R=gpurun_out/$TAG
mkdir -p $R profiles
B="python3 bench.py --no-cpu-baseline --no-fp32-line --serial-streams"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/stats -- $B --steps 5 --warmup 2 > $R/bench_serial_line.json 2> $R/bench_serial.err
cp $R/stats/*/*kernel_stats.csv $R/bench_serial_kernel_stats.csv; rm -rf $R/stats
echo "[profile] stats done"
P="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-fp32-line --serial-streams"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/pmc_fetch -- $P > /dev/null 2> $R/pmc_fetch.err
echo "[profile] fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/pmc_write -- $P > /dev/null 2> $R/pmc_write.err
echo "[profile] write done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/pmc_mfma -- $P > /dev/null 2> $R/pmc_mfma.err
echo "[profile] mfma done"
python3 tools/pmc_traffic.py $R/pmc_fetch/*/*counter_collection.csv $R/pmc_write/*/*counter_collection.csv $R/pmc_hbm_traffic.json > /dev/null
python3 tools/pmc_mfma.py $R/pmc_mfma/*/*counter_collection.csv $R/pmc_mfma/*/*kernel_trace.csv $R/pmc_hbm_traffic.json $R/pmc_utilisation.json > /dev/null
rm -rf $R/pmc_fetch $R/pmc_write $R/pmc_mfma
cp $R/pmc_hbm_traffic.json profiles/${TAG}_pmc_hbm_traffic.json   # bench.py prints the traffic figure from here, with this pass's commit and date
bash tools/gemm_insitu.sh $R/insitu > /dev/null 2>&1; cp $R/insitu/table.txt $R/gemm_insitu_table.txt; cp $R/insitu/by_grid.txt $R/kernels_by_grid.txt
echo "[profile] insitu done"
bash tools/gemm_traffic_insitu.sh $R/gtraffic > /dev/null 2>&1; cp $R/gtraffic/table.txt $R/gemm_traffic_by_shape.txt
echo "[profile] gemm traffic done"
bash tools/kernel_pmc.sh $R/attn_sq tools/attn_bench.py 256 4 > /dev/null 2>&1; cp $R/attn_sq/summary.txt $R/attn_sq_counters.txt
python3 tools/attn_bench.py 256,128 20 > $R/attn_bench.txt 2>&1
echo "[profile] attention counters done"
bash tools/trace_overlap.sh $R/overlap > /dev/null 2>&1; cp $R/overlap/gaps.txt $R/two_stream_overlap.txt
python3 bench.py --steps 20 --warmup 5 > $R/bench_default_line.json 2> $R/bench_default.err
echo "[profile] default bench done"
python3 bench.py --modalities language,image --steps 20 --warmup 5 --no-fp32-line > $R/bench_config1_line.json 2> $R/bench_config1.err
python3 bench.py --modalities video --batch 16 --steps 20 --warmup 5 --no-fp32-line > $R/bench_config4_line.json 2> $R/bench_config4.err
python3 bench.py --missing 0.3 --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-line > $R/bench_config3_line.json 2> $R/bench_config3.err
rm -rf $R/insitu $R/gtraffic $R/attn_sq $R/overlap
ls -la $R
