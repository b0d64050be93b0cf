#!/bin/bash
# additive ablation of the table-fed int8 kernel: start from the MFMAs alone and add the LDS traffic kind by kind, with and
# without the MFMAs.  Builds (CPU side):  bash tools/i8g_ablate2.sh build ; run (GPU box): bash tools/i8g_ablate2.sh
cd "$(dirname "$0")/.."
ALL="-DTXM_G_NO_DMA -DTXM_G_NO_AREAD -DTXM_G_NO_TRREAD -DTXM_G_NO_PRODUCE -DTXM_G_NO_SLICE -DTXM_G_NO_AHEAD"
declare -A V
V[a0_mfma_only]="$ALL"
V[a1_plus_aread]="${ALL/-DTXM_G_NO_AREAD/}"
V[a2_plus_trread]="-DTXM_G_NO_DMA -DTXM_G_NO_PRODUCE -DTXM_G_NO_SLICE -DTXM_G_NO_AHEAD"
V[a3_plus_store]="-DTXM_G_NO_DMA -DTXM_G_NO_SLICE -DTXM_G_NO_AHEAD"
V[a4_plus_ahead]="-DTXM_G_NO_DMA -DTXM_G_NO_SLICE"
V[a5_plus_slice]="-DTXM_G_NO_DMA"
V[a6_full]=""
NAMES="a0_mfma_only a1_plus_aread a2_plus_trread a3_plus_store a4_plus_ahead a5_plus_slice a6_full"
if [ "$1" = build ]; then
  for n in $NAMES; do
    ( bash tools/build_variant.sh tools/build/libtxmom_h_$n.so txm_resample_i8g.hip -DTXM_G_ONLY03 ${V[$n]} $EXTRA_ALL >/dev/null 2>&1
      bash tools/build_variant.sh tools/build/libtxmom_h_${n}_nomfma.so txm_resample_i8g.hip -DTXM_G_ONLY03 -DTXM_T_NO_MFMA ${V[$n]} $EXTRA_ALL >/dev/null 2>&1 ) &
  done
  wait; ls tools/build/libtxmom_h_*.so | wc -l
  exit 0
fi
export TXM_KPATH=int8_table
for n in $NAMES; do
  for sfx in "" _nomfma; do
    export TXM_LIBRARY=$PWD/tools/build/libtxmom_h_$n$sfx.so
    timeout -k 10 200 python tools/ab_kernel.py ${N:-1e8} ${NREP:-1000} 2 2>/dev/null | tail -1
  done
done
