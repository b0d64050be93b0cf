#!/bin/bash
# per-kernel times of BASELINE config 5's batched bootstrap call (GPU box):  bash tools/c5_trace.sh <tag> [library]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${1:-c5}; LIB=${2:-thermoextrap_amd/csrc/libtxmom.so}
D=gpurun_out/${TAG}_trace
rm -rf $D
TXM_LIBRARY=$LIB timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $D -o t --output-format csv -- python3 tools/narrow_time.py c5 5 > $D.log 2>&1 || { echo trace failed; tail -5 $D.log; exit 1; }
python3 tools/top_kernels.py $D 8
