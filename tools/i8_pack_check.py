"""int8 kernel on narrow states (the quad-sharing variant) against the FP64 kernel and, at small size,
the oracle:  python tools/i8_pack_check.py"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

from thermoextrap_amd import engine as eng

torch.manual_seed(0)
worst = 0.0
for C in (1, 3, 8, 9, 13, 16):
    for order in (1, 2, 3, 4, 5, 6, 7):
        for weighted in (False, True):
            N, nrep = 300_000 + 1024 * C + 77, 70
            g = torch.Generator("cuda").manual_seed(C * 100 + order)
            u = torch.empty(N, dtype=torch.float64, device="cuda").normal_(174.85, 5.31, generator=g)
            x = torch.empty((N, C), dtype=torch.float64, device="cuda").normal_(0, 1, generator=g) + 1e-3 * u[:, None]
            w = torch.empty(N, dtype=torch.float64, device="cuda").uniform_(0.2, 3.0, generator=g) if weighted else None
            s = eng.DeviceSampler(5 + C, nrep, N)
            with eng.forced_path("int8"):
                assert eng.resample_path(N, C, nrep, order) == "int8"
                a = eng.resample_vals(x, u, order, sampler=s, w=w)
            with eng.forced_path("fp64"):
                b = eng.resample_vals(x, u, order, sampler=s, w=w)
            K = order + 1
            sc = torch.empty((C, 2, K), dtype=torch.float64, device="cuda")
            for k in range(K):
                sc[:, 0, k] = u.std() ** k
                sc[:, 1, k] = x.std(dim=0) * u.std() ** k
            err = ((a - b).abs() / (b.abs() + sc[None])).max().item()
            worst = max(worst, err)
            flag = "" if err < 2e-12 else "   <-- FAIL"
            print(f"C={C} order={order} weighted={weighted}: max scaled |int8 - fp64| = {err:.2e}{flag}", flush=True)
print("worst", worst)
