#!/bin/bash
# per-kernel times of whole BASELINE config 5 steps (bench.py --config c5; GPU box):  bash tools/c5_step_trace.sh <tag>
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${1:-c5step}
D=gpurun_out/${TAG}_trace
rm -rf $D
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $D -o t --output-format csv -- python3 bench.py --config c5 --steps 10 --warmup 2 > $D.log 2>&1 || { echo trace failed; tail -5 $D.log; exit 1; }
python3 tools/top_kernels.py $D 30
