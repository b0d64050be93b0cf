#!/usr/bin/env python
"""Time a FIRST bootstrap of a data object (pre-pass included) at several shapes:  TXM_LIBRARY=<variant.so> python tools/cold_time.py"""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import thermoextrap_amd as txa
from thermoextrap_amd import engine
from bench import make_data
txa.require_gpu(0)
tag = os.path.basename(os.environ.get("TXM_LIBRARY", "default"))
for N, C, nrep, order in ((100_000_000, 32, 1000, 4), (10_000_000, 8, 200, 4), (1_000_000, 32, 100, 4), (1_000_000, 4, 100, 3), (300_000, 32, 200, 4), (3_000_000, 16, 100, 2)):
    x, u = make_data(N, C, 5, torch)
    s = engine.DeviceSampler(0, nrep, N)
    o = torch.empty((nrep, C, 2, order + 1), dtype=torch.float64, device="cuda")
    fn = lambda: engine.resample_vals(x, u, order, sampler=s, out=o, prep=engine.ResamplePrep())
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort()
    print(f"{tag:26s} cold call N={N:.0e} C={C} nrep={nrep} order={order}: median {ts[2]:8.3f} ms (min {ts[0]:.3f})", flush=True)
    del x, u, s, o
