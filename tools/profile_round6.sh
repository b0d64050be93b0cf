#!/bin/bash
# round-6 evidence (GPU box):  bash tools/profile_round6.sh <tag>
#   1. rocprofv3 kernel stats + FETCH/WRITE passes of the default bench command (tools/profile_round.sh)
#   2. the same trace for config 4
#   3. the bench lines: north star (its `traffic` = the SUM over the call's kernels from step 1's PMC file), configs 2-5
#   4. SQ counter passes: the table kernel at the north star; the narrow launches of configs 2 and 5 (tools/narrow_pmc.sh, before step 3)
#   5. both int8 kernels over the orders (tools/profile_shapes.py) and what every rank of a 2 / 4 / 8-rank run does (tools/scaling_shapes.py)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${1:-r06a}
mkdir -p gpurun_out
bash tools/profile_round.sh $TAG || echo "profile_round failed"
rm -rf gpurun_out/${TAG}_c4_trace
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_c4_trace --output-format csv -- python3 bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_c4_trace.log 2>&1 || echo "c4 trace failed"
python3 - $TAG <<'PY'
import csv, glob, sys
tag = sys.argv[1]
fs = glob.glob(f"gpurun_out/{tag}_c4_trace/*/*kernel_stats.csv")
if fs:
    rows = [r for r in csv.DictReader(open(fs[0])) if "txm::" in r["Name"]]
    with open(f"profiles/{tag}_c4_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f); w.writerow(["Kernel", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
        for r in rows:
            w.writerow([r["Name"].replace("void ", "").split("(")[0], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["Percentage"]])
PY
# narrow states: SQ counters of the contraction launches and FETCH / WRITE of every kernel of the call (configs 2 and 5) -- before the
# bench lines, which read the call's traffic from the summary
bash tools/narrow_pmc.sh ${TAG}_pmc_narrow > gpurun_out/${TAG}_pmc_narrow.log 2>&1 || echo "narrow pmc failed"
python3 bench.py --steps 10 --warmup 2 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || exit 1
for c in c2 c4 c3 c5; do
  python3 bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_$c.json 2>> gpurun_out/${TAG}_bench.err || echo "bench $c failed"
done
# config 4's FETCH / WRITE passes and its bench line again, now with the call's traffic (three passes over one table + the second matrix)
bash tools/profile_c4_traffic.sh $TAG > gpurun_out/${TAG}_c4_traffic.log 2>&1 || echo "c4 traffic failed"
KREGEX=resample_i8g_kernel PMC_ORDER=4 PMC_TAG=${TAG}_pmc_i8g PMC_N=1e8 PMC_NREP=1000 bash tools/i8_pmc.sh 1e8 1000 > gpurun_out/${TAG}_pmc_g.log 2>&1 || echo "pmc g failed"
timeout -k 10 900 python3 tools/profile_shapes.py 1e8 > gpurun_out/${TAG}_shapes.jsonl 2> gpurun_out/${TAG}_shapes.err || echo "shapes failed"
timeout -k 10 600 python3 tools/scaling_shapes.py 1e8 > gpurun_out/${TAG}_scaling_shapes.jsonl 2> gpurun_out/${TAG}_scaling.err || echo "scaling shapes failed"
cp gpurun_out/${TAG}_bench*.json gpurun_out/${TAG}_shapes.jsonl gpurun_out/${TAG}_scaling_shapes.jsonl profiles/ 2>/dev/null
cp profiles/${TAG}* gpurun_out/ 2>/dev/null
tail -3 gpurun_out/${TAG}_pmc_g.log
cut -c1-500 gpurun_out/${TAG}_bench.json
cat gpurun_out/${TAG}_scaling_shapes.jsonl
