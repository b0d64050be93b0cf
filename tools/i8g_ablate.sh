#!/bin/bash
# ablation of the table-fed int8 kernel (GPU box): variant builds made by tools/build_variant.sh (-DTXM_G_NO_* / -DTXM_T_NO_MFMA);
# order 2 = one pass of three row sets (N, NREP, ORDER from the environment).  Results are wrong by construction; times only.
cd "$(dirname "$0")/.."
export TXM_KPATH=int8_table
N=${N:-1e8}; NREP=${NREP:-1000}; ORDER=${ORDER:-2}
for L in default ${VARIANTS:-NO_DMA NO_BARRIER NO_PRODUCE NO_MFMA}; do
  if [ "$L" = default ]; then unset TXM_LIBRARY; else export TXM_LIBRARY=$PWD/tools/build/libtxmom_g_$L.so; fi
  timeout -k 10 200 python tools/ab_kernel.py $N $NREP $ORDER 2>/dev/null | tail -1
done
