#!/usr/bin/env bash
# SQ counter passes over one micro-benchmark command (default: the attention bench at the video tower's shape); per-kernel
# per-dispatch averages are printed by tools/kernel_pmc.py.  Usage on the GPU box: bash tools/kernel_pmc.sh OUTDIR [python args...]
set -uo pipefail
OUT=${1:-gpurun_out/attn_pmc}; shift || true
ARGS=("$@"); [ ${#ARGS[@]} -eq 0 ] && ARGS=(tools/attn_bench.py 256 4)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export HIP_FORCE_DEV_KERNARG=1   # (bench.py sets it itself, but under rocprofv3 the profiler initialises HIP before python starts)
mkdir -p $OUT
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/p$i -- python3 "${ARGS[@]}" > $OUT/p$i.out 2> $OUT/p$i.err || { echo "pass $i failed"; tail -5 $OUT/p$i.err; exit 1; }
done
python3 tools/kernel_pmc.py $OUT/p*/*/*counter_collection.csv > $OUT/summary.txt
rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4
cat $OUT/summary.txt
