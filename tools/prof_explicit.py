#!/usr/bin/env python
"""Diagnostic: bootstrap kernel with an explicit freq table (no Philox stage) to
separate the cost of the fused sampler from the MFMA contraction."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import thermoextrap_amd as txa
from thermoextrap_amd import engine
from bench import make_data

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
nrep = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
C, order = 32, 4
txa.require_gpu(0)
x, u = make_data(N, C, 1000, torch)
s = engine.DeviceSampler(0, nrep, N)
freq = s.freq()
out = torch.empty((nrep, C, 2, order + 1), dtype=torch.float64, device="cuda")
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
fl = 2.0 * N * nrep * (order + 1) * (C + 1)
ta = t(lambda: engine.resample_vals(x, u, order, sampler=s, out=out))
tb = t(lambda: engine.resample_vals(x, u, order, freq=freq, out=out))
print(f"N={N} nrep={nrep}: fused sampler {ta:.2f} ms = {fl/ta/1e9:.1f} TF ; explicit freq {tb:.2f} ms = {fl/tb/1e9:.1f} TF")
