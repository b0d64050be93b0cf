#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of every kernel of BASELINE config 4's step (order 6 + the volume callback's <dx/dq> as a second matrix), so that
# its bench line carries the call's HBM traffic like the north star's (GPU box):  bash tools/profile_c4_traffic.sh <tag>
#   -> profiles/<tag>_c4_traffic.json (tools/collect_profiles.py), then the c4 bench line again -> profiles/<tag>_bench_c4.json
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${1:-r06k}
T=${TAG}_c4
CMD="python3 bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline"
rm -rf gpurun_out/${T}_trace gpurun_out/${T}_pmc_fetch gpurun_out/${T}_pmc_write
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/${T}_trace --output-format csv -- $CMD > gpurun_out/${T}_trace.log 2>&1 || { echo "c4 trace failed"; exit 1; }
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/${T}_pmc_fetch --output-format csv -- $CMD > gpurun_out/${T}_pmc_fetch.log 2>&1 || { echo "c4 fetch failed"; exit 1; }
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/${T}_pmc_write --output-format csv -- $CMD > gpurun_out/${T}_pmc_write.log 2>&1 || { echo "c4 write failed"; exit 1; }
python3 tools/collect_profiles.py $T n_samp=1e8 n_obs=32 order=6 nrep=1000 > gpurun_out/${T}_collect.log 2>&1
python3 bench.py --config c4 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_c4.json 2>> gpurun_out/${TAG}_bench.err || echo "bench c4 failed"
cp profiles/${T}_traffic.json profiles/${T}_kernel_stats.csv gpurun_out/ 2>/dev/null
cut -c1-600 gpurun_out/${TAG}_bench_c4.json
