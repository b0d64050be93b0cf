#!/usr/bin/env python
"""Sixteen-wave int8 kernel (txm_resample_i8w.hip) against the eight-wave one (TXM_I8W=0) bit for bit, and against the
FP64 kernel on the same sampler draw; timing of both:  python tools/i8w_check.py [N] [nrep] [order] [C] [weighted]"""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import thermoextrap_amd as txa
from thermoextrap_amd import engine
from bench import make_data
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
nrep = int(sys.argv[2]) if len(sys.argv) > 2 else 128
order = int(sys.argv[3]) if len(sys.argv) > 3 else 4
C = int(sys.argv[4]) if len(sys.argv) > 4 else 32
weighted = len(sys.argv) > 5 and sys.argv[5] == "1"
txa.require_gpu(0)
x, u = make_data(N, C, 1000, torch)
w = (0.25 + torch.rand(N, dtype=torch.float64, device="cuda")) if weighted else None
s = engine.DeviceSampler(0, nrep, N)
K = order + 1
res = {}
for name, path, env in (("w16", "int8", "1"), ("w8", "int8", "0"), ("fp64", "fp64", "0")):
    os.environ["TXM_I8W"] = env
    with engine.forced_path(path):
        r = engine.resample_vals(x, u, order, sampler=s, w=w)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); r = engine.resample_vals(x, u, order, sampler=s, w=w); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        res[name] = r.clone()
        print(f"{name}: {sorted(ts)[1]:9.3f} ms  info={engine.resample_info()}", flush=True)
sx, su = x[: 1 << 20].std(dim=0), u[: 1 << 20].std()
sc = torch.empty((C, 2, K), dtype=torch.float64, device="cuda")
for b in range(K):
    sc[:, 0, b] = su**b
    sc[:, 1, b] = sx * su**b
nb = (res["w16"] != res["w8"]).sum().item()
print(f"N={N} nrep={nrep} order={order} C={C} w={weighted}: w16 != w8 in {nb} of {res['w16'].numel()} entries")
for nm in ("w16", "w8"):
    d = ((res[nm] - res["fp64"]).abs() / (res["fp64"].abs() + sc[None]))
    print(f"  max scaled |{nm} - fp64|: {d.max().item():.3e}")
if nb:
    d = (res["w16"] - res["w8"]).abs() / (res["w8"].abs() + sc[None])
    print("  max scaled |w16 - w8|:", d.max().item(), "at", torch.nonzero(d == d.max())[0].tolist(), "per (xmom, umom):", d.amax(dim=(0, 1)).tolist())
