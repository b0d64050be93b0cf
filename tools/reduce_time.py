#!/usr/bin/env python
"""Time the HBM-bound reduce (txm_reduce_vals) at the north-star shape:  TXM_LIBRARY=<variant.so> python tools/reduce_time.py [N] [C] [order]"""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import thermoextrap_amd as txa
from thermoextrap_amd import engine
from bench import make_data
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
C = int(sys.argv[2]) if len(sys.argv) > 2 else 32
order = int(sys.argv[3]) if len(sys.argv) > 3 else 4
txa.require_gpu(0)
x, u = make_data(N, C, 3, torch)
engine.reduce_vals(x, u, order); torch.cuda.synchronize()
ts = []
for _ in range(15):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); engine.reduce_vals(x, u, order); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ts.sort()
b = 8.0 * N * (C + 1)
print(f"{os.path.basename(os.environ.get('TXM_LIBRARY', 'default')):30s} reduce N={N:.0e} C={C} order={order}: median {ts[7]:7.3f} ms = {b / ts[7] / 1e9:6.2f} TB/s  (min {ts[0]:.3f})", flush=True)
