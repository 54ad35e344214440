#!/usr/bin/env bash
# Build libmissm_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result ${EXTRA:-}"
OUT=${OUT:-libmissm_hip.so}        # (EXTRA / OUT: experiment builds next to the product library, loaded through MISSM_LIB_PATH)
BUILD=../_build${EXTRA:+_exp}
mkdir -p $BUILD
pids=()
for f in gemm layernorm attention misc lora audio; do
  ( $HIPCC $FLAGS -c $f.hip -o $BUILD/$f.o ${SAVE_TEMPS:+-save-temps=obj} ) &
  pids+=($!)
done
( $HIPCC $FLAGS -c capi.cpp -o $BUILD/capi.o ) &
pids+=($!)
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../$OUT $BUILD/gemm.o $BUILD/layernorm.o $BUILD/attention.o $BUILD/misc.o $BUILD/lora.o $BUILD/audio.o $BUILD/capi.o
echo "built $(realpath ../$OUT)"
