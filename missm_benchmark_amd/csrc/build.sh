#!/usr/bin/env bash
# Build libmissm_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result"
mkdir -p ../_build
pids=()
for f in gemm layernorm attention misc lora audio; do
  ( $HIPCC $FLAGS -c $f.hip -o ../_build/$f.o ${SAVE_TEMPS:+-save-temps=obj} ) &
  pids+=($!)
done
( $HIPCC $FLAGS -c capi.cpp -o ../_build/capi.o ) &
pids+=($!)
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../libmissm_hip.so ../_build/gemm.o ../_build/layernorm.o ../_build/attention.o ../_build/misc.o ../_build/lora.o ../_build/audio.o ../_build/capi.o
echo "built $(realpath ../libmissm_hip.so)"
