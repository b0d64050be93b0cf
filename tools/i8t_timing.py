"""Phase timing of the transposing-read int8 kernel (diagnostic build: -DTXM_I8T_TIMING; GPU box):
TXM_LIBRARY=tools/build/libtxmom_timing.so python tools/i8t_timing.py [N] [order] [nrep]"""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import thermoextrap_amd as txa
from thermoextrap_amd import engine as eng
from bench import make_data
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
order = int(sys.argv[2]) if len(sys.argv) > 2 else 4
nrep = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
C = int(sys.argv[4]) if len(sys.argv) > 4 else 32
txa.require_gpu(0)
x, u = make_data(N, C, 0, torch)
s = eng.DeviceSampler(1, nrep, N)
prep = eng.ResamplePrep()
with eng.forced_path("int8"):
    eng.resample_vals(x, u, order, sampler=s, prep=prep)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); eng.resample_vals(x, u, order, sampler=s, prep=prep); e1.record(); torch.cuda.synchronize()
print(f"call {e0.elapsed_time(e1):.3f} ms", eng.resample_info())
ntiles = -(-N // 1024)
win = 256
WMIN = int(os.environ.get("TXM_WIN_MIN", 256))  # the variant's -DTXM_WIN_MIN
while win > 4 and ntiles < WMIN * win: win //= 4
nwin = -(-ntiles // win)
g0 = -(-(1 + C) * 8 // 256) * 256
off = g0 + nwin * 80 * 8
t = prep.buf[off: off + 2 * 8 * 8 * 8].view(torch.float64).cpu().numpy().reshape(2, 8, 8)
names = ["throttle", "stage+bar+zero", "bar", "fill", "bar", "ksteps", "tail", "flush"]
for b in range(2):
    print("block", ["0", "133"][b])
    for w in range(8):
        tot = t[b, w].sum()
        print(f"  wave {w}: total {tot/1e6:8.2f} Mcyc  " + "  ".join(f"{n} {100*v/tot:4.1f}%" for n, v in zip(names, t[b, w])))
