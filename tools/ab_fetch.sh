#!/bin/bash
# FETCH_SIZE and time of the int8 kernel for several library builds (same box):  bash tools/ab_fetch.sh lib1.so lib2.so ...
cd "$(dirname "$0")/.."
export TMPDIR=/tmp TXM_I8=1
for L in default "$@"; do
  if [ "$L" = default ]; then unset TXM_LIBRARY; else export TXM_LIBRARY=$PWD/$L; fi
  D=gpurun_out/ab_fetch_$(basename $L .so)
  rm -rf ${D}_f ${D}_t
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "resample_i8_kernel" -d ${D}_f -o f --output-format csv -- python3 tools/prof_driver.py 1e8 1000 32 4 1 > ${D}_f.log 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d ${D}_t -o t --output-format csv -- python3 tools/prof_driver.py 1e8 1000 32 4 2 > ${D}_t.log 2>&1
  python3 - "$D" "$L" <<'PY'
import csv, glob, sys
d, l = sys.argv[1:3]
vals = [float(r["Counter_Value"]) for f in glob.glob(d + "_f/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE"]
ms = [float(r["AverageNs"]) / 1e6 for f in glob.glob(d + "_t/**/*kernel_stats.csv", recursive=True) for r in csv.DictReader(open(f)) if "resample_i8_kernel<" in r["Name"]]
print(f"{l:50s} FETCH raw {sum(vals)/max(len(vals),1)/1048576:7.2f} GiB   kernel {ms[0] if ms else float('nan'):7.2f} ms")
PY
done
