#!/usr/bin/env python
"""A/B timing of bootstrap-kernel builds: TXM_LIBRARY=<so> python tools/ab_kernel.py [N] [nrep] [order] [C]"""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import thermoextrap_amd as txa
from thermoextrap_amd import engine
from bench import make_data
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
nrep = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
order = int(sys.argv[3]) if len(sys.argv) > 3 else 4
C = int(sys.argv[4]) if len(sys.argv) > 4 else 32
txa.require_gpu(0)
import _toolenv; _toolenv.apply()
x, u = make_data(N, C, 1000, torch)
s = engine.DeviceSampler(0, nrep, N)
out = torch.empty((nrep, C, 2, order + 1), dtype=torch.float64, device="cuda")
KP = os.environ.get("TXM_KPATH")  # e.g. int8_table / int8_fused: the kernel of the wide int8 path
engine.resample_vals(x, u, order, sampler=s, out=out, path=KP); torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); engine.resample_vals(x, u, order, sampler=s, out=out, path=KP); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
fl = 2.0 * N * nrep * (order + 1) * (C + 1)
t = sorted(ts)[len(ts) // 2]
print(f"{os.environ.get('TXM_LIBRARY','default'):40s} median {t:8.2f} ms  {fl/t/1e9:6.1f} TF  (min {min(ts):.2f})")
