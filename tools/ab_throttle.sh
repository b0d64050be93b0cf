#!/bin/bash
# A/B of the L2-sharing hint (TXM_THROTTLE=0/1) on one bootstrap kernel (GPU box):
#   bash tools/ab_throttle.sh fp64|int8 [N] [nrep]     -> kernel time (rocprofv3 --stats) and FETCH_SIZE per launch
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
PATHSEL=${1:-fp64}; N=${2:-1e8}; NREP=${3:-1000}
if [ "$PATHSEL" = "fp64" ]; then export TXM_I8=0; RX="resample_kernel"; else export TXM_I8=1; RX="resample_i8_kernel"; fi
for T in 0 1; do
  export TXM_THROTTLE=$T
  D=gpurun_out/ab_thr_${PATHSEL}_$T
  rm -rf ${D}_t ${D}_f
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d ${D}_t -o t --output-format csv -- python3 tools/prof_driver.py $N $NREP 32 4 2 > ${D}_t.log 2>&1 || { echo "trace $T failed"; tail -5 ${D}_t.log; exit 1; }
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$RX" -d ${D}_f -o f --output-format csv -- python3 tools/prof_driver.py $N $NREP 32 4 1 > ${D}_f.log 2>&1 || { echo "pmc $T failed"; tail -5 ${D}_f.log; exit 1; }
  python3 - "$D" "$RX" "$T" <<'PY'
import csv, glob, sys
d, rx, t = sys.argv[1:4]
for f in glob.glob(d + "_t/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if rx + "<" in r["Name"]:
            print(f"throttle={t} {r['Name'][:60]:60s} calls={r['Calls']} avg_ms={float(r['AverageNs'])/1e6:.2f}")
vals = []
for f in glob.glob(d + "_f/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            vals.append(float(r["Counter_Value"]))
if vals:
    print(f"throttle={t} FETCH_SIZE per launch = {sum(vals)/len(vals)/1048576:.2f} GiB raw (KiB counter), x2 corrected = {2*sum(vals)/len(vals)/1048576:.2f} GiB")
PY
done
