#!/usr/bin/env python
"""Time WIDE bootstrap calls on the fused int8 kernel (the kernel of calls below the table kernel's thresholds) over series lengths:
   TXM_LIBRARY=<variant.so> python tools/wide_fused_time.py"""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import thermoextrap_amd as txa
from thermoextrap_amd import engine
from bench import make_data

txa.require_gpu(0)
tag = os.path.basename(os.environ.get("TXM_LIBRARY", "default"))


def med(fn, reps=9):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[reps // 2]


for N, C, nrep, order, path in ((1_000_000, 32, 100, 4, None), (1_000_000, 32, 128, 3, None), (1_000_000, 32, 1000, 4, "int8_fused"),
                                (3_000_000, 64, 100, 2, None), (10_000_000, 32, 100, 4, None), (100_000_000, 32, 125, 4, None)):
    x, u = make_data(N, C, 5, torch)
    s = engine.DeviceSampler(0, nrep, N)
    prep = engine.ResamplePrep()
    o = torch.empty((nrep, C, 2, order + 1), dtype=torch.float64, device="cuda")
    m = med(lambda: engine.resample_vals(x, u, order, sampler=s, out=o, prep=prep, path=path))
    print(f"{tag:28s} N={N:>10d} C={C} nrep={nrep:4d} order={order}: {m:8.3f} ms [{engine.resample_info()['kernel']}]", flush=True)
    del x, u, s, o, prep
