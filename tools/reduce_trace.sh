#!/bin/bash
# the reduce kernel's duration as rocprofv3 sees it against HIP events in the same process (GPU box):  bash tools/reduce_trace.sh <library>
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
LIB=${1:-thermoextrap_amd/csrc/libtxmom.so}
D=gpurun_out/reduce_trace_$(basename $LIB .so)
rm -rf $D
TXM_LIBRARY=$LIB timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $D -o t --output-format csv -- python3 tools/reduce_time.py > $D.log 2>&1 || { echo trace failed; tail -3 $D.log; exit 1; }
grep -v amdgpu.ids $D.log | tail -1
python3 tools/top_kernels.py $D 3 | grep reduce_rowmajor
