#!/usr/bin/env python
"""Small driver for rocprofv3: runs the hot-path kernels a few times.
usage: python tools/prof_driver.py [N] [nrep] [C] [order] [iters]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

import thermoextrap_amd as txa
from thermoextrap_amd import engine
from bench import make_data

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
nrep = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
C = int(sys.argv[3]) if len(sys.argv) > 3 else 32
order = int(sys.argv[4]) if len(sys.argv) > 4 else 4
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 2
txa.require_gpu(0)
import _toolenv; _toolenv.apply()
x, u = make_data(N, C, 1000, torch)
state = engine.reduce_vals(x, u, order)
pivot = torch.cat([state[0, 0, 1:2], state[:, 1, 0]]).contiguous()
s = engine.DeviceSampler(0, nrep, N)
out = torch.empty((nrep, C, 2, order + 1), dtype=torch.float64, device="cuda")
for i in range(iters):
    s.draw(100 + i)
    engine.resample_vals(x, u, order, sampler=s, pivot=pivot, out=out)
    engine.reduce_vals(x, u, order)
torch.cuda.synchronize()
print("done", float(out[0, 0, 0, 0]))
