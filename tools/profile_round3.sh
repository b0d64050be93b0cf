#!/bin/bash
# round-3 evidence (GPU box):  bash tools/profile_round3.sh <tag>
#   bench lines of every config, rocprofv3 kernel stats + FETCH/WRITE passes of the default bench command,
#   SQ counter passes of the bootstrap kernel at the north-star size
cd "$(dirname "$0")/.."
TAG=${1:-r03a}
mkdir -p gpurun_out
python3 bench.py --steps 10 --warmup 2 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || exit 1
for c in c2 c4 c3 c5; do
  python3 bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_$c.json 2>> gpurun_out/${TAG}_bench.err || echo "bench $c failed"
done
bash tools/profile_round.sh $TAG || echo "profile_round failed"
PMC_TAG=${TAG}_pmc_i8t PMC_N=1e8 PMC_NREP=1000 bash tools/i8_pmc.sh 1e8 1000 > gpurun_out/${TAG}_pmc.log 2>&1 || echo "pmc failed"
cp gpurun_out/${TAG}_bench*.json profiles/ 2>/dev/null
cp profiles/${TAG}* gpurun_out/ 2>/dev/null
tail -3 gpurun_out/${TAG}_pmc.log
