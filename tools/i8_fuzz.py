"""Randomised parity sweep of the int8 bootstrap kernel against the FP64 kernel on the explicit
frequency table of the same sampler stream (GPU box):  python tools/i8_fuzz.py [cases] [seed]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from thermoextrap_amd import engine as eng
from test_i8_gpu import data, scale, err

ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
for k in range(ncase):
    N = int(rng.choice([1024, 1025, 2047, 3000, 8191, 20000, 65536, 65537, 131072, 300000, 1_000_001]))
    C = int(rng.choice([1, 2, 7, 16, 17, 31, 32, 33, 48, 64, 65]))
    order = int(rng.integers(1, 8))
    nrep = int(rng.choice([1, 2, 63, 64, 65, 100, 128, 200, 257]))
    if N * nrep > 6e7:
        nrep = max(1, int(6e7 // N))
    weighted = bool(rng.integers(0, 2))
    nsamp = 0 if rng.random() < 0.7 else int(N * rng.choice([0.5, 2.0, 3.3]))
    x, u = data(N, C, int(rng.integers(1 << 30)))
    w = (0.1 + torch.rand(N, dtype=torch.float64, device="cuda")) if weighted else None
    s = eng.DeviceSampler(int(rng.integers(1 << 40)), nrep, N, nsamp=nsamp)
    os.environ["TXM_I8"] = "1"
    assert eng.resample_path(N, C, nrep, order) == "int8"
    got = eng.resample_vals(x, u, order, sampler=s, w=w)
    os.environ["TXM_I8"] = "0"
    ref = eng.resample_vals(x, u, order, freq=s.freq(), w=w)
    e = err(got, ref, scale(x, u, order + 1)[None])
    worst = max(worst, e)
    # one rint at 2^-51 of the window maximum per monomial: the highest powers of short series (max / typical
    # ~ 1e4 at order 7, a few hundred draws to average over) sit at 1e-12..1e-11, everything else at 1e-14
    tol = 2e-12 * max(1.0, 4.0 ** (order - 5))
    flag = "" if e < tol else "   <-- FAIL"
    print(f"{k:3d} N={N:8d} C={C:2d} order={order} nrep={nrep:3d} w={int(weighted)} nsamp={nsamp:8d}: {e:.2e}{flag}", flush=True)
    assert torch.isfinite(got).all()
    assert e < tol
print("worst", worst)
