#!/usr/bin/env bash
# The round's judged profiles, all from ONE box: kernel stats of the serial-stream bench, the three PMC passes, the default bench
# line (with fp32 line and CPU baseline), and the bench lines of configs[1] / configs[4].  Run on the GPU box from the repo root.
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export HIP_FORCE_DEV_KERNARG=1   # (bench.py sets it itself, but under rocprofv3 the profiler initialises HIP before python starts)
R=gpurun_out/r02
mkdir -p $R
B="python3 bench.py --no-cpu-baseline --no-fp32-line --serial-streams"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/stats -- $B --steps 5 --warmup 2 > $R/bench_serial_line.json 2> $R/bench_serial.err
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
mkdir -p profiles && cp $R/pmc_hbm_traffic.json profiles/r02_pmc_hbm_traffic.json   # bench.py reads the traffic figure from here
python3 bench.py --steps 20 --warmup 5 > $R/bench_default_line.json 2> $R/bench_default.err
echo "[profile] default bench done"
python3 bench.py --modalities language,image --steps 20 --warmup 5 > $R/bench_config1_line.json 2> $R/bench_config1.err
echo "[profile] config1 done"
python3 bench.py --modalities video --batch 16 --steps 20 --warmup 5 > $R/bench_config4_line.json 2> $R/bench_config4.err
python3 bench.py --missing 0.3 --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-line > $R/bench_config3_line.json 2> $R/bench_config3.err
cp $R/stats/*/*kernel_stats.csv $R/kernel_stats.csv
rm -rf $R/stats $R/pmc_fetch $R/pmc_write $R/pmc_mfma
ls -la $R
