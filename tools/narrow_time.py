#!/usr/bin/env python
"""Time the NARROW-state int8 launches (pre-pass block kept): BASELINE config 2's single call and config 5's batched call.
   TXM_LIBRARY=<variant.so> python tools/narrow_time.py [c2|c5|both] [reps]"""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import thermoextrap_amd as txa
from thermoextrap_amd import engine
from bench import make_data
from tools.bench_states import _state_xu

what = sys.argv[1] if len(sys.argv) > 1 else "both"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
txa.require_gpu(0)
tag = os.path.basename(os.environ.get("TXM_LIBRARY", "default"))
KP = os.environ.get("TXM_KPATH")  # int8_table / int8_fused: the int8 kernel of the narrow launch (None: the library's rule)
tag += f" [{KP}]" if KP else ""


def med(fn):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


out = [f"{tag:34s}"]
if what in ("c2", "both"):
    N, C, order, nrep = 10_000_000, 8, 4, 200
    x, u = make_data(N, C, 1000, torch)
    s = engine.DeviceSampler(0, nrep, N)
    prep = engine.ResamplePrep()
    o = torch.empty((nrep, C, 2, order + 1), dtype=torch.float64, device="cuda")
    m, lo = med(lambda: engine.resample_vals(x, u, order, sampler=s, out=o, prep=prep, path=KP))
    out.append(f"c2 call {m:6.3f} ms (min {lo:6.3f}) [{engine.resample_info()['kernel']}]")
    del x, u
if what in ("c5", "both"):
    S, N, C, order, nrep = 64, 1_000_000, 4, 3, 100
    xs, us = zip(*[_state_xu(torch, s_, N, C) for s_ in range(S)])
    smp = engine.DeviceSampler(1, S * nrep, N)
    prep = engine.ResamplePrep()
    m, lo = med(lambda: engine.resample_vals_batched(list(xs), list(us), order, nrep=nrep, sampler=smp, prep=prep, path=KP))
    out.append(f"c5 batched call {m:6.3f} ms (min {lo:6.3f}) [{engine.batched_info().get('path')}]")
print("  ".join(out), flush=True)
