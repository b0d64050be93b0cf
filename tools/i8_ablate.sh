#!/bin/bash
# A/B builds of the int8 bootstrap kernel with parts switched off (run on the GPU box; the
# TXM_I8_NO_MFMA / _NO_PRODUCE / _NO_FILL macros exist for this script only):
#   bash tools/i8_ablate.sh [N] [nrep] [order]
set -e
cd "$(dirname "$0")/.."
N=${1:-2e7}; NREP=${2:-1000}; ORD=${3:-4}
mkdir -p /tmp/i8ab
CS=thermoextrap_amd/csrc
for v in base NO_MFMA NO_PRODUCE NO_FILL NO_XLOAD "NO_MFMA -DTXM_I8_NO_PRODUCE" "NO_MFMA -DTXM_I8_NO_FILL" "NO_PRODUCE -DTXM_I8_NO_FILL" "NO_MFMA -DTXM_I8_NO_PRODUCE -DTXM_I8_NO_FILL"; do
  tag=$(echo "$v" | tr -d ' ' | tr -c 'A-Za-z0-9_\n' '_')
  def=""; [ "$v" != base ] && def="-DTXM_I8_$v"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-pass-failed $def -c $CS/txm_resample_i8.hip -o /tmp/i8ab/i8_$tag.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/i8ab/lib_$tag.so $CS/build/txm_api.o $CS/build/txm_reduce.o $CS/build/txm_sampler.o \
      $CS/build/txm_small.o $CS/build/txm_resample.o /tmp/i8ab/i8_$tag.o $CS/build/txm_perturb.o
  TXM_I8=1 TXM_LIBRARY=/tmp/i8ab/lib_$tag.so timeout -k 10 200 python tools/ab_kernel.py $N $NREP $ORD
done
