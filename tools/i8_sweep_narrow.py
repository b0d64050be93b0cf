"""Narrow states: when does the int8 path pay (round 4: chunk groups, packed fill)?  python tools/i8_sweep_narrow.py   (GPU box)
Per shape the bootstrap call with the pre-pass block reused (the data object keeps it) and without."""
import sys, torch
sys.path.insert(0, ".")
from thermoextrap_amd import engine as eng
from bench import make_data
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
CS = tuple(int(c) for c in sys.argv[2].split(',')) if len(sys.argv) > 2 else (4, 8, 16)
ORDERS = tuple(int(c) for c in sys.argv[3].split(',')) if len(sys.argv) > 3 else (1, 2, 3, 4)
NREPS = tuple(int(c) for c in sys.argv[4].split(',')) if len(sys.argv) > 4 else (32, 64, 100, 128, 200)
for C in CS:
    x, u = make_data(N, C, 0, torch)
    for order in ORDERS:
        for nrep in NREPS:
            s = eng.DeviceSampler(1, nrep, N)
            t = {}
            for path in ("int8", "fp64"):
                prep = eng.ResamplePrep()
                eng.resample_vals(x, u, order, sampler=s, path=path, prep=prep); torch.cuda.synchronize()
                ts = []
                for _ in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); eng.resample_vals(x, u, order, sampler=s, path=path, prep=prep); e1.record(); torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1))
                t[path] = sorted(ts)[1]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); eng.resample_vals(x, u, order, sampler=s, path="int8"); e1.record(); torch.cuda.synchronize()
            first = e0.elapsed_time(e1)
            auto = eng.resample_path(N, C, nrep, order)
            print(f"C={C:2d} order={order} nrep={nrep:4d}: i8 {t['int8']:7.2f} (first call {first:7.2f})  f64 {t['fp64']:7.2f} ms  "
                  f"ratio {t['fp64']/t['int8']:.2f} / {t['fp64']/first:.2f}   rule: {auto}", flush=True)
