import os, sys, torch, numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from thermoextrap_amd import engine as eng
from test_i8_gpu import data, scale, err
for (N, C, order, nrep) in [(1024, 32, 4, 64), (2048, 32, 4, 64), (40000, 32, 4, 130), (5000, 5, 4, 3), (20000, 32, 2, 100)]:
    x, u = data(N, C, 5)
    s = eng.DeviceSampler(20261003 + N, nrep, N)
    os.environ["TXM_I8"] = "1"
    a = eng.resample_vals(x, u, order, sampler=s).clone()
    b = eng.resample_vals(x, u, order, sampler=s).clone()
    os.environ["TXM_I8"] = "0"
    ref = eng.resample_vals(x, u, order, freq=s.freq()).clone()
    sc = scale(x, u, order + 1)[None]
    d = ((a - b).abs() / (b.abs() + sc))
    print(N, C, order, nrep, "a-vs-b", d.max().item(), "a-vs-ref", err(a, ref, sc), "b-vs-ref", err(b, ref, sc))
    e = ((a - ref).abs() / (ref.abs() + sc))
    idx = torch.nonzero(e > 1e-12)
    print("  n bad", idx.shape[0], "of", e.numel(), "first:", idx[:8].tolist())
    if idx.shape[0]:
        r, c, i, j = idx[0].tolist()
        print("  a", a[r, c].flatten().tolist()); print("  ref", ref[r, c].flatten().tolist())
