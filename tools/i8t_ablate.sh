#!/bin/bash
# ablation of the transposing-read kernel (GPU box): builds made by tools/build_variant.sh with -DTXM_T_NO_*
cd "$(dirname "$0")/.."
export TXM_I8=1
N=${N:-2e7}
for L in default ${VARIANTS:-NO_MFMA NO_PRODUCE NO_TR NO_FILL NO_WRITE NO_LOAD}; do
  if [ "$L" = default ]; then unset TXM_LIBRARY; else export TXM_LIBRARY=$PWD/tools/build/libtxmom_t_$L.so; fi
  timeout -k 10 200 python tools/ab_kernel.py $N 1000 2>/dev/null | tail -1
done
