import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import thermoextrap_amd as txa
from thermoextrap_amd import engine
from bench import make_data
N, nrep, C = 300_000, 64, 32
txa.require_gpu(0)
x, u = make_data(N, C, 1000, torch)
s = engine.DeviceSampler(0, nrep, N)
for order in (0, 1, 4):
    with engine.forced_path("int8"):
        a = engine.resample_vals(x, u, order, sampler=s)
    with engine.forced_path("fp64"):
        b = engine.resample_vals(x, u, order, sampler=s)
    torch.cuda.synchronize()
    print("order", order)
    print(" sum w int8 :", a[:6, 0, 0, 0].tolist())
    print(" sum w fp64 :", b[:6, 0, 0, 0].tolist())
    print(" <x_c> diff by column (rep 0):", ((a[0, :, 1, 0] - b[0, :, 1, 0]).abs() / x.std(dim=0)).tolist())
    print(" <x_0> diff by rep:", ((a[:, 0, 1, 0] - b[:, 0, 1, 0]).abs() / x[:, 0].std())[:16].tolist())
