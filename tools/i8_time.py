import os, sys, time, torch
sys.path.insert(0, ".")
from thermoextrap_amd import engine as eng
from bench import make_data
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
C, order, nrep = 32, 4, int(sys.argv[2]) if len(sys.argv) > 2 else 1000
x, u = make_data(N, C, 0, torch)
s = eng.DeviceSampler(1, nrep, N)
out = {}
for mode in ("1", "0"):
    eng._L().txm_set_resample_path(int(mode))
    r = eng.resample_vals(x, u, order, sampler=s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(2):
        r = eng.resample_vals(x, u, order, sampler=s)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 2
    out[mode] = r.clone()
    flops = 2.0 * N * nrep * (order + 1) * (C + 1)
    print(f"TXM_I8={mode} N={N} nrep={nrep}: {ms:.2f} ms  {flops / ms / 1e9:.1f} TFLOP/s-equivalent", flush=True)
sc = out["0"].abs().amax(dim=0, keepdim=True) * 0 + out["0"].std(dim=0, keepdim=True) + 1e-300
d = (out["1"] - out["0"]).abs()
print("max |i8 - f64| / |f64|:", (d / out["0"].abs().clamp_min(1e-300)).max().item())
