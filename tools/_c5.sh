cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rm -rf gpurun_out/c5_t
rocprofv3 --kernel-trace --stats -d gpurun_out/c5_t -o t --output-format csv -- python3 bench.py --config c5 --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/c5_t.log 2>&1
python3 - <<'PY'
import glob, csv
f = glob.glob("gpurun_out/c5_t/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:40]:
    print("  %-90s calls %5s avg %9.1f us  total %10.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e3))
PY
