#!/usr/bin/env python
"""Top kernels of a rocprofv3 --kernel-trace --stats --output-format csv run:  python tools/top_kernels.py <dir> [n]"""
import csv, glob, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
f = glob.glob(f"{d}/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:n]:
    print(f'{r["Name"][:100]:100s} calls {r["Calls"]:>5s}  avg {float(r["AverageNs"]) / 1e3:10.1f} us  total {float(r["TotalDurationNs"]) / 1e6:9.2f} ms')
