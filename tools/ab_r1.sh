#!/bin/bash
# same-box comparison of the round-1 kernels (tree copied to .r1cmp/) with the current ones:
#   bash tools/ab_r1.sh [N] [nrep]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
N=${1:-1e8}; NREP=${2:-1000}
for P in 0 1; do
  export TXM_I8=$P
  for TREE in .r1cmp . .r1cmp .; do
    D=$PWD/gpurun_out/ab_r1_${P}_$(echo $TREE | tr -d './')x
    rm -rf $D
    ( cd $TREE && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $D -o t --output-format csv -- python3 tools/prof_driver.py $N $NREP 32 4 2 > $D.log 2>&1 )
    python3 - "$D" "$TREE" "$P" <<'PY'
import csv, glob, sys
d, tree, p = sys.argv[1:4]
for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "resample_kernel<" in r["Name"] or "resample_i8_kernel<" in r["Name"] or "window" in r["Name"]:
            print(f"TXM_I8={p} tree={tree:8s} {r['Name'][:58]:58s} calls={r['Calls']} avg_ms={float(r['AverageNs'])/1e6:.2f} max_ms={float(r['MaxNs'])/1e6:.2f}")
PY
  done
done
