#!/bin/bash
# same-box A/B of int8-kernel builds:  bash tools/ab_lib.sh <lib.so> [<lib2.so> ...]   (each vs the default library, twice, interleaved)
cd "$(dirname "$0")/.."
export TXM_I8=1
N=${N:-1e8}; NREP=${NREP:-1000}
for rep in 1 2; do
  for L in default "$@"; do
    if [ "$L" = default ]; then unset TXM_LIBRARY; else export TXM_LIBRARY=$PWD/$L; fi
    timeout -k 10 200 python tools/ab_kernel.py $N $NREP 2>/dev/null | tail -1
  done
done
