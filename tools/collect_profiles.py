#!/usr/bin/env python
"""Condense rocprofv3 output (gpurun_out/<tag>_trace, <tag>_pmc_fetch, <tag>_pmc_write)
into the tracked summaries under profiles/:

  profiles/<tag>_kernel_stats.csv   per-kernel Calls / total / average ns (txm kernels only)
  profiles/<tag>_traffic.json       per-kernel FETCH_SIZE / WRITE_SIZE per launch and the
                                    HBM byte figures derived from them as
                                    /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes

usage: python tools/collect_profiles.py r01
"""
import collections
import csv
import glob
import json
import sys
from pathlib import Path

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = Path(__file__).resolve().parent.parent
out = root / "profiles"
out.mkdir(exist_ok=True)


def short(name):
    n = name.replace("void ", "")
    return n.split("(")[0]


stats = glob.glob(str(root / f"gpurun_out/{tag}_trace/*/*kernel_stats.csv"))
rows = []
if stats:
    for r in csv.DictReader(open(stats[0])):
        if "txm::" in r["Name"]:
            rows.append({"Kernel": short(r["Name"]), "Calls": r["Calls"], "TotalDurationNs": r["TotalDurationNs"],
                         "AverageNs": r["AverageNs"], "MinNs": r["MinNs"], "MaxNs": r["MaxNs"],
                         "Percentage": r["Percentage"]})
    with open(out / f"{tag}_kernel_stats.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)

traffic = {}
for kind, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    files = glob.glob(str(root / f"gpurun_out/{tag}_pmc_{kind}/*/*counter_collection.csv"))
    if not files:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        if "txm::" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        traffic.setdefault(k, {})[counter + "_KiB_per_launch"] = sum(v) / len(v)
        traffic[k][counter + "_launches"] = len(v)
for k, d in traffic.items():
    f = d.get("FETCH_SIZE_KiB_per_launch", 0.0) * 1024
    w = d.get("WRITE_SIZE_KiB_per_launch", 0.0) * 1024
    d["read_bytes_raw"] = f
    # gfx950: FETCH_SIZE counts 128-B requests as 64 B for wide coalesced streams -> x2
    d["read_bytes_corrected_x2"] = 2 * f
    d["write_bytes"] = w
    d["hbm_bytes_per_launch"] = 2 * f + w
workload = {"n_samp": 100_000_000, "n_obs": 32, "order": 4, "nrep": 1000}
for a in sys.argv[2:]:  # e.g. n_samp=1e8 order=6
    k, v = a.split("=")
    workload[k] = int(float(v))
sys.path.insert(0, str(root))
from bench import csrc_sha  # noqa: E402

json.dump({"tag": tag, "workload": workload, "csrc_sha": csrc_sha(),
           "command": "python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline (one rocprofv3 --pmc pass per counter)",
           "note": "FETCH_SIZE x2 correction per MI355X_MICROARCH.md (calibrated for 16 B/lane streams; "
           "the bootstrap kernel's 8 B/lane x loads are uncalibrated, so its figure is an upper bound)",
           "kernels": traffic}, open(out / f"{tag}_traffic.json", "w"), indent=1)
print(open(out / f"{tag}_kernel_stats.csv").read() if rows else "no trace")
print(json.dumps(traffic, indent=1)[:1500])
