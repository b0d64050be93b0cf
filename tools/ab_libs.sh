#!/bin/bash
# same-box A/B of whole-library builds:  bash tools/ab_libs.sh <lib1.so> <lib2.so> ...   (N, NREP, ORDER from the environment)
cd "$(dirname "$0")/.."
export TXM_I8=1
N=${N:-2e7}; NREP=${NREP:-1000}; ORDER=${ORDER:-4}
for rep in 1 2; do
  for L in "$@"; do
    export TXM_LIBRARY=$PWD/$L
    timeout -k 10 300 python tools/ab_kernel.py $N $NREP $ORDER 2>/dev/null | tail -1
  done
done
