#!/bin/bash
# memory-side PMC passes over the bootstrap kernel (GPU box):  bash tools/i8t_mem_pmc.sh [N] [nrep]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp TXM_I8=1
N=${1:-2e7}; NREP=${2:-1000}
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM_RD"; do
  i=$((i+1))
  rm -rf gpurun_out/mem_pmc$i
  timeout -k 10 240 rocprofv3 --pmc $set --kernel-include-regex "${KREGEX:-resample_i8t?_kernel}" -d gpurun_out/mem_pmc$i -o pmc --output-format csv -- \
      python3 tools/prof_driver.py $N $NREP 32 4 1 > gpurun_out/mem_pmc$i.log 2>&1 || { echo "pass $i failed"; tail -5 gpurun_out/mem_pmc$i.log; }
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.OrderedDict()
for f in sorted(glob.glob("gpurun_out/mem_pmc*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(f"{k:32s} {sum(v)/len(v):.4e}  (n={len(v)})")
PY
