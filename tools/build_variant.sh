#!/bin/bash
# Build a variant of libtxmom.so with extra flags on ONE source file (A/B and diagnostic builds):
#   bash tools/build_variant.sh <out.so> <source.hip> <flags...>      e.g.  tools/build/libtxmom_timing.so txm_resample_i8t.hip -DTXM_I8T_TIMING
set -e
cd "$(dirname "$0")/.."
OUT=$1; SRC=$2; shift 2
C=thermoextrap_amd/csrc
mkdir -p tools/build
O=tools/build/$(basename "$OUT" .so)_$(basename "$SRC" .hip).o
EXTRA=""
if [ "$SRC" = txm_sampler.hip ]; then EXTRA="-ffp-contract=off"; fi
if [ "$SRC" = txm_resample_i8t.hip ]; then EXTRA="-mllvm -amdgpu-mfma-vgpr-form -Wno-inline-asm"; fi   # as thermoextrap_amd/_build.py
if [ "$SRC" = txm_resample_i8gn.hip ]; then EXTRA="-mllvm -amdgpu-mfma-vgpr-form -mllvm -greedy-regclass-priority-trumps-globalness=1 -Wno-inline-asm"; fi
if [ "$SRC" = txm_resample_i8g.hip ]; then EXTRA="-mllvm -amdgpu-mfma-vgpr-form -mllvm -greedy-regclass-priority-trumps-globalness=1 -Wno-inline-asm"; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-pass-failed $EXTRA "$@" -c $C/$SRC -o $O
OBJS=""
for f in $(python3 -c "from thermoextrap_amd._build import SOURCES; print(' '.join(s[:-4] for s in SOURCES))"); do
  if [ "$f.hip" = "$SRC" ]; then OBJS="$OBJS $O"; else OBJS="$OBJS $C/build/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJS
echo built $OUT
