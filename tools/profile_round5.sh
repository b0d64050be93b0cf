#!/bin/bash
# round-5 evidence (GPU box):  bash tools/profile_round5.sh <tag>
#   1. rocprofv3 kernel stats + FETCH/WRITE passes of the default bench command (north star: the fused int8 kernel)
#   2. the same trace for config 4 (order 6 + <dx/dq>: the count-table kernel and its generator)
#   3. the bench lines (north star first: `traffic` from step 1's PMC file, same kernel sources), configs 2-5
#   4. SQ counter passes: the fused kernel at the north star, the table kernel at order 6
#   5. both int8 kernels over the orders (tools/profile_shapes.py)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${1:-r05a}
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
    print(open(f"profiles/{tag}_c4_kernel_stats.csv").read())
PY
python3 bench.py --steps 10 --warmup 2 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || exit 1
for c in c2 c4 c3 c5; do
  python3 bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_$c.json 2>> gpurun_out/${TAG}_bench.err || echo "bench $c failed"
done
# (every wide order runs the table kernel by rule now; the fused kernel's counters: a 125-replicate slab of the north star, which it keeps)
PMC_ORDER=${PMC_ORDER_FUSED:-4} PMC_TAG=${TAG}_pmc_i8t PMC_N=1e8 PMC_NREP=125 bash tools/i8_pmc.sh 1e8 125 > gpurun_out/${TAG}_pmc.log 2>&1 || echo "pmc failed"
KREGEX=resample_i8g_kernel PMC_ORDER=${PMC_ORDER_TABLE:-4} PMC_TAG=${TAG}_pmc_i8g PMC_N=1e8 PMC_NREP=1000 bash tools/i8_pmc.sh 1e8 1000 > gpurun_out/${TAG}_pmc_g.log 2>&1 || echo "pmc g failed"
timeout -k 10 900 python3 tools/profile_shapes.py 1e8 > gpurun_out/${TAG}_shapes.jsonl 2> gpurun_out/${TAG}_shapes.err || echo "shapes failed"
cp gpurun_out/${TAG}_bench*.json gpurun_out/${TAG}_shapes.jsonl profiles/ 2>/dev/null
cp profiles/${TAG}* gpurun_out/ 2>/dev/null
tail -3 gpurun_out/${TAG}_pmc.log; tail -3 gpurun_out/${TAG}_pmc_g.log
cut -c1-600 gpurun_out/${TAG}_bench.json
cat gpurun_out/${TAG}_shapes.jsonl
