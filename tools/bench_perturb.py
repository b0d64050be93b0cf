#!/usr/bin/env python
"""Timing of the perturbation kernel at the north-star data size (HBM roofline)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import thermoextrap_amd as txa
from thermoextrap_amd import engine
from bench import make_data
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
C = 32
txa.require_gpu(0)
x, u = make_data(N, C, 1000, torch)
for na in (1, 4, 8):
    da = [0.01 * (k + 1) for k in range(na)]
    engine.perturb(x, u, da); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): engine.perturb(x, u, da)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"perturb N={N:.0e} C={C} n_alpha={na}: {ms:.2f} ms  {8.0*N*(C+1)/ms/1e6:.0f} GB/s ({8.0*N*(C+1)/ms/1e6/80:.1f} % of 8 TB/s)")
