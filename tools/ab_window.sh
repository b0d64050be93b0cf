#!/bin/bash
# same-box A/B of the scaling-window rule on short series (round 6): the product (>= 256 windows where N allows) against
# variants that keep windows longer (-DTXM_WIN_MIN=64 / 16: tools/build_variant.sh <out> txm_resample.hip -DTXM_WIN_MIN=..)
cd "$(dirname "$0")/.."
for round in 1 2; do
  for lib in "" tools/build/libtxmom_win64.so tools/build/libtxmom_win16.so; do
    if [ -n "$lib" ] && [ ! -f "$lib" ]; then continue; fi
    TXM_LIBRARY=${lib:-thermoextrap_amd/csrc/libtxmom.so} python tools/narrow_time.py both 9 || exit 1
  done
done
