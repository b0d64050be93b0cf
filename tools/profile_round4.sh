#!/bin/bash
# round-4 evidence (GPU box):  bash tools/profile_round4.sh <tag>
#   1. rocprofv3 kernel stats + FETCH/WRITE passes of the default bench command (tools/profile_round.sh)
#   2. THEN the bench lines (north star first: its `traffic` is taken from the PMC file of step 1, same kernel sources),
#   3. SQ counter passes of the bootstrap kernel at the north-star size,
#   4. the shapes that had no profile (tools/profile_shapes.py: weighted, replicate slab, orders 0 / 6 (+ second matrix))
cd "$(dirname "$0")/.."
TAG=${1:-r04a}
mkdir -p gpurun_out
bash tools/profile_round.sh $TAG || echo "profile_round failed"
python3 bench.py --steps 10 --warmup 2 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || exit 1
for c in c2 c4 c3 c5; do
  python3 bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_$c.json 2>> gpurun_out/${TAG}_bench.err || echo "bench $c failed"
done
PMC_TAG=${TAG}_pmc_i8t PMC_N=1e8 PMC_NREP=1000 bash tools/i8_pmc.sh 1e8 1000 > gpurun_out/${TAG}_pmc.log 2>&1 || echo "pmc failed"
timeout -k 10 600 python3 tools/profile_shapes.py 1e8 > gpurun_out/${TAG}_shapes.jsonl 2> gpurun_out/${TAG}_shapes.err || echo "shapes failed"
cp gpurun_out/${TAG}_bench*.json gpurun_out/${TAG}_shapes.jsonl profiles/ 2>/dev/null
cp profiles/${TAG}* gpurun_out/ 2>/dev/null
tail -3 gpurun_out/${TAG}_pmc.log
cut -c1-400 gpurun_out/${TAG}_bench.json
cat gpurun_out/${TAG}_shapes.jsonl
