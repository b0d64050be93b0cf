#!/bin/bash
# rocprofv3 evidence for one round (GPU box):  bash tools/profile_round.sh <tag>
#   kernel trace + stats of the default bench command, then separate PMC passes (FETCH_SIZE, WRITE_SIZE)
#   as MI355X_MICROARCH.md prescribes; tools/collect_profiles.py condenses them into profiles/.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${1:-r01b}
CMD="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rm -rf gpurun_out/${TAG}_trace gpurun_out/${TAG}_pmc_fetch gpurun_out/${TAG}_pmc_write
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_trace --output-format csv -- $CMD > gpurun_out/${TAG}_trace.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/${TAG}_pmc_fetch --output-format csv -- $CMD > gpurun_out/${TAG}_pmc_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/${TAG}_pmc_write --output-format csv -- $CMD > gpurun_out/${TAG}_pmc_write.log 2>&1 || exit 1
python3 tools/collect_profiles.py $TAG > gpurun_out/${TAG}_collect.log 2>&1
cp profiles/${TAG}_kernel_stats.csv profiles/${TAG}_traffic.json gpurun_out/ 2>/dev/null
tail -3 gpurun_out/${TAG}_collect.log
