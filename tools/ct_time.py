#!/usr/bin/env python
"""Time the count-table generator alone (txm_sampler_count_table):  TXM_LIBRARY=<variant.so> python tools/ct_time.py [N] [nrep]"""
import ctypes as ct, os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import thermoextrap_amd as txa
from thermoextrap_amd import _lib, engine
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
nrep = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
txa.require_gpu(0)
L = _lib.load()
s = engine.DeviceSampler(0, nrep, N)
tb = torch.empty(L.txm_sampler_count_table_bytes(N, nrep), dtype=torch.uint8, device="cuda")
def gen():
    _lib.check(L.txm_sampler_count_table(ct.byref(s.spec), engine._ptr(s.counts), 0, nrep, engine._ptr(tb), engine._stream()), "count_table")
gen(); torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gen(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
ts.sort()
print(f"{os.path.basename(os.environ.get('TXM_LIBRARY', 'default')):34s} count table N={N:.0e} nrep={nrep}: median {ts[2]:7.2f} ms  min {ts[0]:7.2f}  ({N * (-(-nrep // 128) * 128) / ts[2] / 1e9:.2f} TB/s written)", flush=True)
