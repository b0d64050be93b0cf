#!/bin/bash
# PMC passes over the FP64 bootstrap kernel (GPU box):  bash tools/fp64_pmc.sh [N] [nrep] [C] [order]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp TXM_I8=0
N=${1:-1e7}; NREP=${2:-200}; C=${3:-8}; ORD=${4:-4}
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
           "SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rm -rf gpurun_out/fp64_pmc$i
  timeout -k 10 240 rocprofv3 --pmc $set --kernel-include-regex "resample_kernel" -d gpurun_out/fp64_pmc$i -o pmc --output-format csv -- \
      python3 tools/prof_driver.py $N $NREP $C $ORD 1 > gpurun_out/fp64_pmc$i.log 2>&1 || { echo "pass $i failed"; tail -5 gpurun_out/fp64_pmc$i.log; }
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.OrderedDict()
for f in sorted(glob.glob("gpurun_out/fp64_pmc*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "resample_kernel" in r["Kernel_Name"]:
            agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(f"{k:32s} {sum(v)/len(v):.4e}  (n={len(v)})")
PY
