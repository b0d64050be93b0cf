#!/usr/bin/env python
"""Timing of the bootstrap shapes beside the north star (round 5: both int8 kernels where the rule could take either; round 4: the shapes that had no profile: the weighted north star, a rep0-offset
125-replicate slab (what one of 8 ranks runs in `bench.py --mode replicas`), orders 6 (+ second matrix: config 4's call) and 0.
One JSON line per shape:  python tools/profile_shapes.py [N]"""
import json, sys
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
w = 0.25 + torch.rand(N, dtype=torch.float64, device="cuda")


def timed(label, order, nrep, rep0=0, weights=None, y=None, path=None):
    s = engine.DeviceSampler(3, nrep, N, rep0=rep0)
    prep = engine.ResamplePrep()
    kw = dict(sampler=s, w=weights, prep=prep, path=path)
    if y is not None:
        kw["y"] = y
    engine.resample_vals(x, u, order, **kw)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); engine.resample_vals(x, u, order, **kw); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    t = sorted(ts)[1]
    K = order + 1
    info = engine.resample_info()
    if info.get("kernel") == "int8_table":
        # table kernel: 128 replicates per workgroup, 4 quarters x 8 quads = 32 MFMAs per row set and k-step, u-row in the dead byte
        mf = 32 * (K + (1 if y is not None else 0))
        ngrp = -(-nrep // 128)
    else:
        # fused kernel: executed MFMAs per workgroup k-step: 16 JN x tiles + 2 ceil(JN / 4) u-row tiles per pass (+ 16 for a second matrix)
        def tiles(jn):
            return 16 * jn + 2 * ((jn + 3) // 4)
        mf = {1: tiles(1), 2: tiles(2), 3: tiles(3), 4: tiles(4), 5: tiles(5), 6: tiles(3) + tiles(3), 7: tiles(4) + tiles(3), 8: tiles(4) + tiles(4)}[K]
        if y is not None:
            mf += 16
        ngrp = -(-nrep // 64)
    ops = mf * ngrp * (-(-N // 1024)) * 32 * 65536.0
    print(json.dumps({"shape": label, "n_samp": N, "n_obs": C, "order": order, "nrep": nrep, "rep0": rep0, "weighted": weights is not None,
                      "second_matrix": y is not None, "ms_per_call": round(t, 3), "info": info,
                      "executed_int8_TOPs": round(ops / t / 1e9, 1), "frac_of_5000": round(ops / t / 1e9 / 5000.0, 4)}), flush=True)


timed("north star", 4, 1000)
timed("north star, weighted", 4, 1000, weights=w)
timed("replicate slab 125 @ rep0=375 (1 of 8 ranks, --mode replicas)", 4, 125, rep0=375)
timed("north star on the count-table kernel (the rule keeps order 4 on the fused one: a tie)", 4, 1000, path="int8_table")
for o in (0, 1, 2, 3, 5, 6, 7):
    timed(f"order {o}", o, 1000)
    timed(f"order {o}, fused kernel forced", o, 1000, path="int8_fused")
if N <= 100_000_000:
    y = x * 0.5 + 1.0
    timed("order 6 + second matrix (config 4's call)", 6, 1000, y=y)
    timed("order 6 + second matrix, fused kernel forced", 6, 1000, y=y, path="int8_fused")
    timed("order 4 + second matrix", 4, 1000, y=y)
