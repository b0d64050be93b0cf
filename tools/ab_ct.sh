#!/bin/bash
# same-box A/B of the count-table generator (round 6): the product against variants given as arguments, alone at several shapes
# bash tools/ab_ct.sh lib1.so lib2.so ...
cd "$(dirname "$0")/.."
for round in 1 2; do
  for lib in "$@"; do
    for shape in "1e7 200" "1e7 1000" "1e6 384" "3e6 128" "1e8 1000"; do
      TXM_LIBRARY=$lib python3 tools/ct_time.py $shape 2>&1 | grep -v amdgpu.ids || exit 1
    done
  done
done
