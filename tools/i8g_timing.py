"""Phase clocks of the table-fed int8 kernel (diagnostic build: -DTXM_G_TIMING; GPU box):
TXM_LIBRARY=tools/build/libtxmom_gtiming.so python tools/i8g_timing.py [N] [order] [nrep]
(single-pass orders -- 0, 1, 2 -- show one pass; with more passes the last one's clocks are read)"""
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
names = ["prologue+flush", "tail of step 3", "block-start DMA issue", "step bodies", "x wait", "staging: raw read wait", "dma wait + barrier", "staging rest"]
for b in range(2):
    print("workgroup", ["0", "1064"][b])
    for w in range(8):
        tot = t[b, w].sum()
        print(f"  wave {w}: total {tot/1e6:8.2f} Mcyc  " + "  ".join(f"{n} {100*v/max(tot,1):4.1f}%" for n, v in zip(names, t[b, w])))
