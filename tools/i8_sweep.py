"""When does the int8 path pay?  python tools/i8_sweep.py   (GPU box)"""
import os, sys, torch
sys.path.insert(0, ".")
from thermoextrap_amd import engine as eng
from bench import make_data
N = 10_000_000
for C in (20, 32):
    x, u = make_data(N, C, 0, torch)
    for order in (2, 3, 4):
        for nrep in (64, 128, 200, 300, 400):
            s = eng.DeviceSampler(1, nrep, N)
            t = {}
            for mode in ("1", "0"):
                eng._L().txm_set_resample_path(int(mode))
                eng.resample_vals(x, u, order, sampler=s); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); eng.resample_vals(x, u, order, sampler=s); e1.record(); torch.cuda.synchronize()
                t[mode] = e0.elapsed_time(e1)
            print(f"C={C:2d} order={order} nrep={nrep:4d}: i8 {t['1']:8.2f} ms  f64 {t['0']:8.2f} ms  ratio {t['0']/t['1']:.2f}", flush=True)
