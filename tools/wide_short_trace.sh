#!/bin/bash
# per-kernel times of a WIDE bootstrap call on a short / medium series (GPU box):  bash tools/wide_short_trace.sh <N> [nrep] [order] [library]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
N=${1:-1e6}; NREP=${2:-1000}; ORD=${3:-4}; LIB=${4:-thermoextrap_amd/csrc/libtxmom.so}
D=gpurun_out/wide_${N}_trace
rm -rf $D
TXM_LIBRARY=$LIB timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $D -o t --output-format csv -- python3 tools/prof_driver.py $N $NREP 32 $ORD 3 > $D.log 2>&1 || { echo trace failed; tail -5 $D.log; exit 1; }
python3 tools/top_kernels.py $D 9
