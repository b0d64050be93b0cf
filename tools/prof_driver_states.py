#!/usr/bin/env python
"""Small driver for rocprofv3: the BATCHED bootstrap of S narrow state points (BASELINE config 5's launch) a few times.
usage: python tools/prof_driver_states.py [S] [N] [C] [order] [nrep] [iters]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

import thermoextrap_amd as txa
from thermoextrap_amd import engine
from tools.bench_states import _state_xu

S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
C = int(sys.argv[3]) if len(sys.argv) > 3 else 4
order = int(sys.argv[4]) if len(sys.argv) > 4 else 3
nrep = int(sys.argv[5]) if len(sys.argv) > 5 else 100
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 2
txa.require_gpu(0)
xs, us = zip(*[_state_xu(torch, s, N, C) for s in range(S)])
smp = engine.DeviceSampler(1, S * nrep, N)
prep = engine.ResamplePrep()
for i in range(iters + 1):   # (the first call fills the pre-pass block; the others are the launch of a step)
    smp.draw(100 + i)
    out = engine.resample_vals_batched(list(xs), list(us), order, nrep=nrep, sampler=smp, prep=prep)
torch.cuda.synchronize()
print("done", engine.batched_info(), float(out[0, 0, 0, 0, 0]))
