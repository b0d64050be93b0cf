"""Effective shader clock inside the table-fed int8 kernel (diagnostic build: -DTXM_G_CLOCKS; GPU box):
TXM_LIBRARY=tools/build/libtxmom_h_clocks.so python tools/i8g_clocks.py [N] [order] [nrep]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import thermoextrap_amd as txa
from thermoextrap_amd import engine as eng
from bench import make_data
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
order = int(sys.argv[2]) if len(sys.argv) > 2 else 2
nrep = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
C = 32
txa.require_gpu(0)
x, u = make_data(N, C, 0, torch)
s = eng.DeviceSampler(1, nrep, N)
prep = eng.ResamplePrep()
for _ in range(3):
    eng.resample_vals(x, u, order, sampler=s, prep=prep, path="int8_table")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); eng.resample_vals(x, u, order, sampler=s, prep=prep, path="int8_table"); e1.record(); torch.cuda.synchronize()
print(f"call {e0.elapsed_time(e1):.2f} ms", eng.resample_info())
ntiles = -(-N // 1024)
win = 256
while win > 4 and ntiles < 256 * win: win //= 4
nwin = -(-ntiles // win)
g0 = 2 * (-(-(1 + C) * 8 // 256) * 256)
off = g0 + nwin * 80 * 8
t = prep.buf[off: off + 2 * 8 * 8 * 8].view(torch.float64).cpu().numpy().reshape(2, 8, 8)
for b in range(2):
    for w in (0, 7):
        cyc, ref = t[b, w, 0], t[b, w, 1]
        print(f"workgroup {['0','1064'][b]} wave {w}: {cyc/1e6:.2f} M shader cycles in {ref/100:.1f} us of the 100 MHz reference clock -> {cyc/max(ref,1)*100:.0f} MHz")
