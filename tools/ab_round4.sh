#!/bin/bash
# same-box A/B of the default library and variant builds (tools/build/libtxmom_<NAME>.so):  bash tools/ab_round4.sh NAME...
# (N, NREP, ORDER from the environment; every library is timed twice, interleaved)
cd "$(dirname "$0")/.."
export TXM_I8=1
N=${N:-2e7}; NREP=${NREP:-1000}; ORDER=${ORDER:-4}; NOBS=${NOBS:-32}
for rep in 1 2; do
  unset TXM_LIBRARY
  timeout -k 10 300 python tools/ab_kernel.py $N $NREP $ORDER $NOBS 2>/dev/null | tail -1
  for L in "$@"; do
    export TXM_LIBRARY=$PWD/tools/build/libtxmom_$L.so
    timeout -k 10 300 python tools/ab_kernel.py $N $NREP $ORDER $NOBS 2>/dev/null | tail -1
  done
done
