#!/usr/bin/env python
"""One-GPU timings of what EVERY rank of an N-rank run does, so that the first 8-GPU SCALE record can be held against a
prediction (round-5 verdict item 7).  North-star shape (N = 1e8, N_obs = 32, order 4), one JSON line per item:

  --mode replicas (strong scaling): rank r of n bootstraps nrep / n replicates at stream offset rep0 = r nrep / n of the SAME
      state point -- slabs of 1000, 500, 250, 125 replicates at the first, a middle and the last offset; the kernel the
      dispatch rule takes, ms per call, the sampler's tile-count kernel, derivatives + D2H.
  --mode states   (weak scaling): every rank runs the one-GPU step on its own state point -- the 1000-replicate line.
  sharded_reduce: N / n rows per rank -- pivot estimate, per-rank power sums (the HBM-bound reduce), sums -> state.

python tools/scaling_shapes.py [N] > profiles/rNN_scaling_shapes.jsonl"""
import json, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import thermoextrap_amd as txa
from thermoextrap_amd import engine
from bench import make_data

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
C, order, NREP = 32, 4, 1000
txa.require_gpu(0)
x, u = make_data(N, C, 1000, torch)


def ev(fn, reps=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


prep = engine.ResamplePrep()
for world in (1, 2, 4, 8):
    n = NREP // world
    for r in sorted({0, world // 2, world - 1}):
        rep0 = r * n
        t_s = ev(lambda: engine.DeviceSampler(7, n, N, rep0=rep0))
        s = engine.DeviceSampler(7, n, N, rep0=rep0)
        out = torch.empty((n, C, 2, order + 1), dtype=torch.float64, device="cuda")
        t_b = ev(lambda: engine.resample_vals(x, u, order, sampler=s, out=out, prep=prep))
        info = engine.resample_info()
        rec = {"item": "replica slab", "world": world, "rank": r, "rep0": rep0, "nrep": n, "kernel": info.get("kernel"),
               "prep_reused": info.get("prep_reused"), "sampler_ms": round(t_s, 3), "bootstrap_call_ms": round(t_b, 3)}
        for forced in ("int8_fused", "int8_table"):
            rec[forced + "_ms"] = round(ev(lambda: engine.resample_vals(x, u, order, sampler=s, out=out, prep=engine.ResamplePrep(), path=forced), 2), 3)
        print(json.dumps(rec), flush=True)
for world in (1, 2, 4, 8):
    m = N // world
    xs, us = x[:m], u[:m]
    t_p = ev(lambda: engine.reduce_pivot(xs, us), 5)
    piv = engine.reduce_pivot(xs, us)
    t_r = ev(lambda: engine.reduce_sums(xs, us, order, piv), 5)
    sums = engine.reduce_sums(xs, us, order, piv)
    stack = sums.unsqueeze(0).repeat(world, 1, 1, 1)
    t_f = ev(lambda: engine.sums_to_state(stack, piv), 5)
    print(json.dumps({"item": "sharded_reduce phase", "world": world, "rows_per_rank": m, "pivot_ms": round(t_p, 4), "sums_ms": round(t_r, 4),
                      "sums_GBs": round(8.0 * m * (C + 1) / t_r / 1e6, 1), "sums_to_state_ms": round(t_f, 4),
                      "all_gather_bytes_per_rank": int(sums.numel() * 8)}), flush=True)
