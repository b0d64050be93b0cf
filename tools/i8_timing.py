"""Phase timing of the int8 kernel (build with -DTXM_I8_TIMING; GPU box):
TXM_LIBRARY=<timing build> python tools/i8_timing.py [N] [order]"""
import os, sys, ctypes as ct
import torch
sys.path.insert(0, ".")
os.environ["TXM_I8"] = "1"
os.environ["TXM_THROTTLE"] = os.environ.get("TXM_THROTTLE", "1")
from thermoextrap_amd import engine as eng, _lib
from bench import make_data
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
C, order, nrep = 32, (int(sys.argv[2]) if len(sys.argv) > 2 else 4), 1000
x, u = make_data(N, C, 0, torch)
s = eng.DeviceSampler(1, nrep, N)
L = _lib.load()
nbytes = L.txm_resample_vals_ws_bytes(N, C, nrep, order)
with eng.forced_path("int8"):
    eng.resample_vals(x, u, order, sampler=s)
torch.cuda.synchronize()
ws = eng.workspace(nbytes)
# locate the window table: recompute the plan offsets like plan_i8 (n_chunks=16, nrep_pad=1024 for this shape)
ntiles = -(-N // 1024); K = order + 1
n_rbg = -(-nrep // 64); nrep_pad = n_rbg * 64
nc = max(8, 256 // n_rbg // 8 * 8)
win = 256
while win > 4 and -(-ntiles // win) < 8 * nc: win //= 4
nwin = -(-ntiles // win)
tpc = -(-nwin // nc) * win
n_chunks = -(-(-(-ntiles // tpc)) // 8) * 8
al = lambda v: -(-v // 256) * 256
off_px = al((1 + C) * 8)
off_pu = off_px + al(n_chunks * 7 * nrep_pad * 32 * K * 8)
off_wt = off_pu + al(n_chunks * 7 * nrep_pad * K * 8)
t = ws[off_wt + nwin * 80 * 8: off_wt + nwin * 80 * 8 + 2 * 8 * 8 * 8].view(torch.float64).cpu().numpy().reshape(2, 8, 8)
names = ["zero", "barrier(fill)", "fill", "mfma", "produce", "barrier(step)", "flush", "tile setup"]
for b in range(2):
    print("block", ["0", "133"][b])
    for w in range(8):
        tot = t[b, w].sum()
        print(f"  wave {w}: total {tot/1e6:8.2f} Mcyc  " + "  ".join(f"{n} {100*v/tot:4.1f}%" for n, v in zip(names, t[b, w])))
