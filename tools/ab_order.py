#!/usr/bin/env python
"""Time the device-sampler bootstrap at a given order on both kernels: python tools/ab_order.py [order] [N] [nrep] [C]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import thermoextrap_amd as txa
from thermoextrap_amd import engine
from bench import make_data
order = int(sys.argv[1]) if len(sys.argv) > 1 else 0
N = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100_000_000
nrep = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
C = int(sys.argv[4]) if len(sys.argv) > 4 else 32
txa.require_gpu(0)
x, u = make_data(N, C, 1000, torch)
s = engine.DeviceSampler(0, nrep, N)
res = {}
for path in ("int8", "fp64"):
    with engine.forced_path(path):
        if engine.resample_path(N, C, nrep, order) != path:
            print(path, "not available"); continue
        out = engine.resample_vals(x, u, order, sampler=s); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); out = engine.resample_vals(x, u, order, sampler=s); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        res[path] = out.clone()
        print(f"order {order} N={N:.0e} nrep={nrep} C={C} {path}: {sorted(ts)[1]:.2f} ms", flush=True)
if len(res) == 2:
    sc = res["fp64"].std(dim=0, keepdim=True) + res["fp64"].abs().mean(dim=0, keepdim=True)
    print("max scaled |int8 - fp64|:", ((res["int8"] - res["fp64"]).abs() / sc).max().item())
