#!/bin/bash
# L2 (TCC) counters of the int8 kernel for several library builds:  bash tools/ab_tcc.sh lib1.so ...
cd "$(dirname "$0")/.."
export TMPDIR=/tmp TXM_I8=1
for L in default "$@"; do
  if [ "$L" = default ]; then unset TXM_LIBRARY; else export TXM_LIBRARY=$PWD/$L; fi
  i=0
  for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum TCC_EA0_RDREQ_DRAM_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"; do
    i=$((i+1)); D=gpurun_out/ab_tcc_$(basename $L .so)_$i; rm -rf $D
    timeout -k 10 300 rocprofv3 --pmc $set --kernel-include-regex "resample_i8_kernel" -d $D -o p --output-format csv -- python3 tools/prof_driver.py 1e8 1000 32 4 1 > $D.log 2>&1
  done
  python3 - "$L" <<'PY'
import csv, glob, sys, collections
l = sys.argv[1]
import os
b = os.path.basename(l).replace(".so", "")
agg = collections.OrderedDict()
for f in sorted(glob.glob(f"gpurun_out/ab_tcc_{b}_*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
print(l, {k: f"{sum(v)/len(v):.3e}" for k, v in agg.items()})
PY
done
