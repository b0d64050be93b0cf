#!/bin/bash
# same-box A/B (round 6): the chunk groups' flush -- serial through group 0 (before) against summed in the count tile and
# written out by all eight waves -- and on top of it longer windows on short series (-DTXM_WIN_MIN=16)
cd "$(dirname "$0")/.."
for round in 1 2; do
  for lib in thermoextrap_amd/csrc/libtxmom.so tools/build/libtxmom_flush.so tools/build/libtxmom_win16.so tools/build/libtxmom_flush_win16.so; do
    [ -f "$lib" ] || continue
    TXM_KPATH=int8_fused TXM_LIBRARY=$lib python tools/narrow_time.py both 9 || exit 1
  done
done
