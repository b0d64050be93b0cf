#!/bin/bash
# kernel times, HBM fetch and SQ counters of the table-fed int8 path (GPU box):  bash tools/i8g_prof.sh <tag> [N] [nrep] [order]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${1:-r05a}; N=${2:-1e8}; NREP=${3:-1000}; ORD=${4:-4}
D=gpurun_out/${TAG}_i8g
rm -rf ${D}_t ${D}_f
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d ${D}_t -o t --output-format csv -- python3 tools/prof_driver.py $N $NREP 32 $ORD 2 > ${D}_t.log 2>&1 || { echo trace failed; tail -5 ${D}_t.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --kernel-include-regex "resample_i8g_kernel|count_table_kernel" -d ${D}_f -o f --output-format csv -- python3 tools/prof_driver.py $N $NREP 32 $ORD 1 > ${D}_f.log 2>&1 || { echo fetch failed; tail -5 ${D}_f.log; }
python3 - "$D" <<'PY'
import csv, glob, sys, collections
D = sys.argv[1]
for f in glob.glob(D + "_t/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r["TotalDurationNs"]) > 2e5:
            print(f"{r['Name'][:90]:90s} calls={r['Calls']:>3s} avg_ms={float(r['AverageNs'])/1e6:8.3f}")
agg = collections.defaultdict(list)
for f in glob.glob(D + "_f/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    # FETCH_SIZE / WRITE_SIZE: kilobytes; gfx950: FETCH_SIZE reports half the bytes of wide streaming reads (MI355X_MICROARCH.md)
    print(f"{k:60s} {c:12s} per launch {sum(v)/len(v)/1e6:9.2f} GB raw" + ("  (x2 = %.1f GB corrected)" % (2 * sum(v) / len(v) / 1e6) if c == "FETCH_SIZE" else ""))
PY
KREGEX=resample_i8g_kernel PMC_TAG=${TAG}_pmc_i8g PMC_N=$N PMC_NREP=$NREP bash tools/i8_pmc.sh $N $NREP > gpurun_out/${TAG}_pmc_i8g.log 2>&1 || echo "pmc failed"
tail -40 gpurun_out/${TAG}_pmc_i8g.log
