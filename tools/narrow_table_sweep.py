#!/usr/bin/env python
"""Narrow states (C <= 16): the fused kernel (draws in place, 64 replicates a workgroup) against the table-fed one
(txm_resample_i8gn.hip: generator + contraction, 128 replicates a workgroup), pre-pass block kept, median of 5 calls.
   python tools/narrow_table_sweep.py [N ...]          -> one line per shape: fused ms, table ms, fused / table"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import thermoextrap_amd as txa
from thermoextrap_amd import engine
from bench import make_data

txa.require_gpu(0)
Ns = [int(float(v)) for v in sys.argv[1:]] or [1_000_000, 10_000_000]


def med(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[reps // 2]


from thermoextrap_amd import _build
print(f"# csrc_sha {_build.csrc_sha()}", flush=True)
for N in Ns:
    for C in (2, 4, 5, 8, 12, 16):
        x, u = make_data(N, C, 7, torch)
        for order in (1, 2, 3, 4, 6):
            for nrep in (64, 100, 128, 200, 256, 512, 1000):
                s = engine.DeviceSampler(0, nrep, N)
                o = torch.empty((nrep, C, 2, order + 1), dtype=torch.float64, device="cuda")
                r = {}
                for p in ("int8_fused", "int8_table"):
                    prep = engine.ResamplePrep()
                    r[p] = med(lambda: engine.resample_vals(x, u, order, sampler=s, out=o, prep=prep, path=p))
                    ran = engine.resample_info()["kernel"]
                if ran != "int8_table":  # (operands the table kernel does not take: a row shorter than a column quad, an odd pitch)
                    print(f"N={N:>9d} C={C:2d} order={order} nrep={nrep:4d}: fused {r['int8_fused']:8.3f} ms  table not applicable", flush=True)
                    continue
                print(f"N={N:>9d} C={C:2d} order={order} nrep={nrep:4d}: fused {r['int8_fused']:8.3f} ms  table {r['int8_table']:8.3f} ms  "
                      f"fused/table {r['int8_fused'] / r['int8_table']:.2f}", flush=True)
                del s, o
        del x, u
