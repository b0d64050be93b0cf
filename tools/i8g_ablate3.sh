#!/bin/bash
# second ablation series (see i8g_ablate2.sh): where do the slicing's vector instructions and the DMA pieces lose their time?
cd "$(dirname "$0")/.."
declare -A V
V[b1_nodma]="-DTXM_G_NO_DMA"
V[b2_nodma_nooverlay]="-DTXM_G_NO_DMA -DTXM_G_NO_OVERLAY"
V[b3_nodma_noahead]="-DTXM_G_NO_DMA -DTXM_G_NO_AHEAD"
V[b4_nodma_nopin]="-DTXM_G_NO_DMA -DTXM_G_NO_PIN"
V[b5_nodma_nooverlay_noahead]="-DTXM_G_NO_DMA -DTXM_G_NO_OVERLAY -DTXM_G_NO_AHEAD"
V[c0_full]=""
V[c1_nopin]="-DTXM_G_NO_PIN"
V[c2_noxdma]="-DTXM_G_NO_XDMA"
V[c3_noadma]="-DTXM_G_NO_ADMA"
V[c4_nobarrier]="-DTXM_G_NO_BARRIER"
NAMES=${NAMES:-"b1_nodma b2_nodma_nooverlay b3_nodma_noahead b4_nodma_nopin b5_nodma_nooverlay_noahead c0_full c1_nopin c2_noxdma c3_noadma c4_nobarrier"}
if [ "$1" = build ]; then
  for n in $NAMES; do
    bash tools/build_variant.sh tools/build/libtxmom_h_$n.so txm_resample_i8g.hip -DTXM_G_ONLY03 ${V[$n]} >/dev/null 2>&1 &
  done
  wait; ls tools/build/libtxmom_h_[bc]*.so | wc -l
  exit 0
fi
export TXM_KPATH=int8_table
for n in $NAMES; do
  export TXM_LIBRARY=$PWD/tools/build/libtxmom_h_$n.so
  timeout -k 10 200 python tools/ab_kernel.py ${N:-1e8} ${NREP:-1000} 2 2>/dev/null | tail -1
done
