#!/bin/bash
# HBM / L2 traffic of the table-fed int8 kernel and its generator, one counter set per rocprofv3 pass (GPU box):
#   bash tools/i8g_traffic.sh [N] [nrep] [order]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
N=${1:-1e8}; NREP=${2:-1000}; ORDER=${3:-2}
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  rm -rf gpurun_out/i8g_tr$i
  timeout -k 10 240 rocprofv3 --pmc $set --kernel-include-regex "resample_i8g_kernel|count_table_kernel" -d gpurun_out/i8g_tr$i -o pmc --output-format csv -- \
      python3 tools/prof_driver.py $N $NREP 32 $ORDER 1 > gpurun_out/i8g_tr$i.log 2>&1 || { echo "pass $i ($set) failed"; tail -3 gpurun_out/i8g_tr$i.log; }
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.OrderedDict()
for f in sorted(glob.glob("gpurun_out/i8g_tr*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0].replace("void ", "")[:60], r["Counter_Name"])
        agg.setdefault(k, []).append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(f"{k[0]:62s} {k[1]:28s} mean {sum(v)/len(v):.4e}  (n={len(v)})")
PY
