#!/bin/bash
# Counters of the NARROW-state int8 launches (round-5 verdict item 3: no PMC summary existed for them):
#   c2  one state, N = 1e7, C = 8, order 4, nrep = 200          (tools/prof_driver.py)
#   c5  64 states batched, N = 1e6, C = 4, order 3, nrep = 100  (tools/prof_driver_states.py)
# Four SQ passes + FETCH_SIZE + WRITE_SIZE (separate --pmc passes, --kernel-include-regex on the contraction kernel), summary in
# profiles/<tag>.json.   bash tools/narrow_pmc.sh <tag>
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${1:-r06_pmc_narrow}
RX="resample_i8[a-z]*_kernel"
# FETCH_SIZE / WRITE_SIZE: every kernel of the bootstrap call (generator, contraction, listed FP64 launch, finalize, info word), so that
# the call's traffic can be summed as bench.call_traffic does for the north star
RXALL="resample_i8[a-z]*_kernel|count_table_kernel|resample_finalize|i8_info_kernel|resample_kernel"
declare -A CMD
CMD[c2]="tools/prof_driver.py 1e7 200 8 4 1"
CMD[c5]="tools/prof_driver_states.py 64 1e6 4 3 100 1"
SETS=("SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"
      "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"
      "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD"
      "SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_I8"
      "FETCH_SIZE" "WRITE_SIZE")
for cfg in c2 c5; do
  i=0
  for set in "${SETS[@]}"; do
    i=$((i+1))
    D=gpurun_out/${TAG}_${cfg}_$i
    rm -rf $D
    R="$RX"; if [ "$set" = FETCH_SIZE ] || [ "$set" = WRITE_SIZE ]; then R="$RXALL"; fi
    timeout -k 10 300 rocprofv3 --pmc $set --kernel-include-regex "$R" -d $D -o pmc --output-format csv -- \
        python3 ${CMD[$cfg]} > $D.log 2>&1 || { echo "$cfg pass $i failed"; tail -5 $D.log; }
  done
  rm -rf gpurun_out/${TAG}_${cfg}_t
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_${cfg}_t -o t --output-format csv -- python3 ${CMD[$cfg]} > gpurun_out/${TAG}_${cfg}_t.log 2>&1 || echo "$cfg trace failed"
done
python3 - "$TAG" <<'PY'
import csv, glob, json, sys, collections
sys.path.insert(0, ".")
from bench import csrc_sha
tag = sys.argv[1]
out = {"tag": tag, "csrc_sha": csrc_sha(),
       "command": "bash tools/narrow_pmc.sh (six rocprofv3 --pmc passes per configuration, one process each; counters summed over the chip, "
                  "averaged over the launches of the contraction kernel; FETCH_SIZE / WRITE_SIZE in KiB, HBM bytes = 2 x FETCH + WRITE as "
                  "MI355X_MICROARCH.md prescribes for gfx950)",
       "configs": {}}
shapes = {"c2": {"states": 1, "n_samp": 10_000_000, "n_obs": 8, "order": 4, "nrep": 200},
          "c5": {"states": 64, "n_samp": 1_000_000, "n_obs": 4, "order": 3, "nrep": 100}}
for cfg in ("c2", "c5"):
    agg = collections.OrderedDict()
    names = set()
    perk = {}   # FETCH_SIZE / WRITE_SIZE of every kernel of the call: name -> counter -> [values]
    for f in sorted(glob.glob(f"gpurun_out/{tag}_{cfg}_[0-9]/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"].replace("void ", "").split("(")[0]
            if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                perk.setdefault(kn, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            if "resample_i8" not in kn:
                continue
            agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            names.add(kn)
    c = {k: sum(v) / len(v) for k, v in agg.items()}
    kt = {}
    for kn, d_ in perk.items():
        fk = sum(d_.get("FETCH_SIZE", [0.0])) / max(len(d_.get("FETCH_SIZE", [0.0])), 1) * 1024
        wk = sum(d_.get("WRITE_SIZE", [0.0])) / max(len(d_.get("WRITE_SIZE", [0.0])), 1) * 1024
        kt[kn] = {"read_bytes_corrected_x2": 2 * fk, "write_bytes": wk, "hbm_bytes_per_launch": 2 * fk + wk,
                  "launches": len(d_.get("FETCH_SIZE", []))}
    ms = None
    for f in glob.glob(f"gpurun_out/{tag}_{cfg}_t/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "resample_i8" in r["Name"]:
                ms = float(r["AverageNs"]) / 1e6
    d = {"workload": shapes[cfg], "kernels": sorted(names), "kernel_avg_ms": ms, "launches_averaged": {k: len(v) for k, v in agg.items()}, "counters": c}
    if c:
        fetch, write = c.get("FETCH_SIZE", 0) * 1024, c.get("WRITE_SIZE", 0) * 1024
        w = shapes[cfg]
        alg = 8.0 * w["states"] * w["n_samp"] * (w["n_obs"] + 1)
        d["hbm_bytes_per_launch"] = 2 * fetch + write
        d["algorithmic_bytes"] = alg
        d["traffic_ratio"] = (2 * fetch + write) / alg
        # the whole call (prep block reused): the sum bench.call_traffic forms from the per-kernel figures
        import bench
        info_kernel = "int8_table" if any("i8gn" in n or "i8g_" in n for n in names) else "int8_fused"
        tot, per = bench.call_traffic(kt, info_kernel, w["n_obs"])
        d["kernels_traffic"] = kt
        d["call_kernel"] = info_kernel
        d["call_hbm_bytes"] = tot
        d["call_traffic_ratio"] = tot / alg
        d["call_kernels"] = per
        wave = max(c.get("SQ_WAVE_CYCLES", 1), 1)
        d["derived"] = {
            "valu_instructions_per_mfma": c.get("SQ_INSTS_VALU", 0) / max(c.get("SQ_INSTS_MFMA", 1), 1),
            "wave_cycles_waiting_fraction (SQ_WAIT_ANY / SQ_WAVE_CYCLES)": c.get("SQ_WAIT_ANY", 0) / wave,
            "wave_cycles_issue_stalled_fraction (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES)": c.get("SQ_WAIT_INST_ANY", 0) / wave,
            "mfma_busy_cycles_per_simd": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024,
            "valu_issue_cycles_per_simd (4 x SQ_ACTIVE_INST_VALU / 1024)": 4.0 * c.get("SQ_ACTIVE_INST_VALU", 0) / 1024,
            "lds_idx_active_cycles_per_cu": c.get("SQ_LDS_IDX_ACTIVE", 0) / 256,
            "lds_bank_conflict_cycles_per_cu": c.get("SQ_LDS_BANK_CONFLICT", 0) / 256,
            "busy_cycles_per_se (SQ_BUSY_CYCLES / 32)": c.get("SQ_BUSY_CYCLES", 0) / 32,
        }
    out["configs"][cfg] = d
json.dump(out, open(f"profiles/{tag}.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
PY
cp profiles/${TAG}.json gpurun_out/ 2>/dev/null
