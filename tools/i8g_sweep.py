#!/usr/bin/env python
"""Table-fed int8 kernel ("int8") against the kernel that draws in place ("int8_fused"): ms per bootstrap call (pre-pass block kept),
wide shapes.   python tools/i8g_sweep.py [big]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import thermoextrap_amd as txa
from thermoextrap_amd import engine
from bench import make_data

txa.require_gpu(0)
big = len(sys.argv) > 1
shapes = [(100_000_000, 32, 1000, o, False) for o in (1, 2, 3, 5, 7)] if big else []
shapes += [(10_000_000, 32, 1000, 4, False), (10_000_000, 32, 200, 4, False), (10_000_000, 32, 200, 2, False), (10_000_000, 32, 200, 6, False),
           (10_000_000, 32, 100, 4, False), (10_000_000, 32, 64, 4, False), (10_000_000, 32, 130, 3, False), (10_000_000, 24, 256, 3, False),
           (10_000_000, 64, 256, 3, False), (1_000_000, 32, 1000, 4, False), (1_000_000, 32, 256, 2, False), (300_000, 32, 512, 4, False),
           (10_000_000, 32, 256, 4, True), (10_000_000, 32, 256, 6, True), (10_000_000, 32, 256, 1, True)]
last = None
for (N, C, nrep, order, withy) in shapes:
    if last != (N, C):
        x, u = make_data(N, C, 3, torch)
        y = x * 0.5 + 1.0
        last = (N, C)
    s = engine.DeviceSampler(1, nrep, N)
    t = {}
    for path in ("int8_fused", "int8_table"):
        prep = engine.ResamplePrep()
        kw = dict(sampler=s, path=path, prep=prep, y=y if withy else None)
        engine.resample_vals(x, u, order, **kw); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); engine.resample_vals(x, u, order, **kw); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        t[path] = min(ts)
    print(f"N={N:>9d} C={C} nrep={nrep:4d} order={order} y={int(withy)}: fused {t['int8_fused']:8.2f} ms   table {t['int8_table']:8.2f} ms   ratio {t['int8_fused'] / t['int8_table']:.2f}", flush=True)
