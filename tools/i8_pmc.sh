#!/bin/bash
# PMC passes over the int8 bootstrap kernel (GPU box):  bash tools/i8_pmc.sh [N] [nrep]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp TXM_I8=${TXM_I8:-1}
N=${1:-2e7}; NREP=${2:-1000}
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD" \
           "SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  rm -rf gpurun_out/i8_pmc$i
  timeout -k 10 240 rocprofv3 --pmc $set --kernel-include-regex "${KREGEX:-resample_i8t?_kernel}" -d gpurun_out/i8_pmc$i -o pmc --output-format csv -- \
      python3 tools/prof_driver.py $N $NREP 32 ${PMC_ORDER:-4} 1 > gpurun_out/i8_pmc$i.log 2>&1 || { echo "pass $i failed"; tail -5 gpurun_out/i8_pmc$i.log; }
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.OrderedDict()
for f in sorted(glob.glob("gpurun_out/i8_pmc*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(f"{k:32s} {sum(v)/len(v):.4e}  (n={len(v)})")
import json, os, sys
sys.path.insert(0, ".")
from bench import csrc_sha
c = {k: sum(v) / len(v) for k, v in agg.items()}
tag = os.environ.get("PMC_TAG")
if tag and c:
    q = 4.0  # SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md)
    d = {"tag": tag, "kernel_regex": os.environ.get("KREGEX", "resample_i8t?_kernel"), "csrc_sha": csrc_sha(),
         "workload": {"n_samp": int(float(os.environ.get("PMC_N", "2e7"))), "n_obs": 32, "order": int(os.environ.get("PMC_ORDER", "4")), "nrep": int(os.environ.get("PMC_NREP", "1000"))},
         "command": "bash tools/i8_pmc.sh N NREP (four rocprofv3 --pmc passes of tools/prof_driver.py, one launch each; counters summed over the chip)",
         "counters": c,
         "derived": {
             "valu_instructions_per_mfma": c.get("SQ_INSTS_VALU", 0) / max(c.get("SQ_INSTS_MFMA", 1), 1),
             "wave_cycles_waiting_fraction (SQ_WAIT_ANY / SQ_WAVE_CYCLES)": c.get("SQ_WAIT_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1),
             "wave_cycles_issue_stalled_fraction (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES)": c.get("SQ_WAIT_INST_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1),
             "lds_bank_conflict_cycles_per_cu": c.get("SQ_LDS_BANK_CONFLICT", 0) / 256,
             "lds_bank_conflict_ms_per_cu_at_2.4GHz": c.get("SQ_LDS_BANK_CONFLICT", 0) / 256 / 2.4e6,
             "lds_idx_active_cycles_per_cu": c.get("SQ_LDS_IDX_ACTIVE", 0) / 256,
             "mfma_busy_cycles_per_simd": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024,
             "valu_issue_cycles_per_simd (4 x SQ_ACTIVE_INST_VALU / 1024)": q * c.get("SQ_ACTIVE_INST_VALU", 0) / 1024,
         }}
    json.dump(d, open(f"profiles/{tag}.json", "w"), indent=1)
    print("wrote", f"profiles/{tag}.json")
PY
