#!/usr/bin/env python
"""Batched bootstrap of S narrow states: the int8 path (state on a grid axis of every kernel) against S single int8 calls
bit for bit, against the batched FP64 kernel, and the timing of both:  python tools/batched_i8_check.py [S] [N] [C] [order] [nrep]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import thermoextrap_amd as txa
from thermoextrap_amd import engine
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
C = int(sys.argv[3]) if len(sys.argv) > 3 else 4
order = int(sys.argv[4]) if len(sys.argv) > 4 else 3
nrep = int(sys.argv[5]) if len(sys.argv) > 5 else 100
outlier = len(sys.argv) > 6 and sys.argv[6] == "1"
txa.require_gpu(0)
g = torch.Generator(device="cuda").manual_seed(7)
xs, us = [], []
for s in range(S):
    u = 170.0 + s + 5.0 * torch.randn(N, generator=g, dtype=torch.float64, device="cuda")
    x = 2.0 + 0.01 * u[:, None] + 0.4 * torch.randn(N, C, generator=g, dtype=torch.float64, device="cuda")
    if outlier and s == S // 2:
        u[N // 3] += 5.0e4  # one window of ONE state goes to the FP64 kernel
    xs.append(x); us.append(u)
smp = engine.DeviceSampler(11, S * nrep, N)
K = order + 1
res = {}
for path in ("int8", "fp64"):
    with engine.forced_path(path):
        r = engine.resample_vals_batched(xs, us, order, nrep=nrep, sampler=smp)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); r = engine.resample_vals_batched(xs, us, order, nrep=nrep, sampler=smp); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        res[path] = r.clone()
        print(f"batched {path}: {sorted(ts)[2]:9.3f} ms", flush=True)
        if path == "int8":  # the pre-pass block kept by the caller
            prep = engine.ResamplePrep()
            ts = []
            for _ in range(6):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); r = engine.resample_vals_batched(xs, us, order, nrep=nrep, sampler=smp, prep=prep); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            print(f"batched int8, pre-pass block reused: {sorted(ts[1:])[2]:9.3f} ms (hits {prep.hits}, misses {prep.misses}); equal: {torch.equal(r, res['int8'])}", flush=True)
r = engine.resample_vals_batched(xs, us, order, nrep=nrep, sampler=smp)
print("default dispatch == int8:", torch.equal(r, res["int8"]), " == fp64:", torch.equal(r, res["fp64"]))
nb = 0
with engine.forced_path("int8"):
    for s in range(S):
        one = engine.resample_vals(xs[s], us[s], order, sampler=engine.DeviceSampler(11, nrep, N, rep0=s * nrep))
        nb += (one != res["int8"][s]).sum().item()
        if s == S // 2:
            print("state", s, "single call:", engine.resample_info())
print(f"batched int8 != serial int8 in {nb} of {res['int8'].numel()} entries")
sc = torch.empty((S, 1, C, 2, K), dtype=torch.float64, device="cuda")
for s in range(S):
    for b in range(K):
        sc[s, 0, :, 0, b] = us[s][:100000].std() ** b
        sc[s, 0, :, 1, b] = xs[s][:100000].std(dim=0) * us[s][:100000].std() ** b
d = (res["int8"] - res["fp64"]).abs() / (res["fp64"].abs() + sc)
print(f"max scaled |int8 - fp64| = {d.max().item():.3e}")
