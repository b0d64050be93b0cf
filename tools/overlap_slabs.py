#!/usr/bin/env python
"""Experiment: one bootstrap call as replicate slabs on SEVERAL streams, so that the count-table generator of one slab (HBM-write
bound) overlaps the contraction passes of another (latency bound).  Rows [a, b) of a call are bit for bit the call at rep0 = a.
   python tools/overlap_slabs.py [N] [nrep] [order]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import thermoextrap_amd as txa
from thermoextrap_amd import engine
from bench import make_data
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
nrep = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
order = int(sys.argv[3]) if len(sys.argv) > 3 else 4
C = 32
txa.require_gpu(0)
x, u = make_data(N, C, 1000, torch)
s = engine.DeviceSampler(0, nrep, N)
prep = engine.ResamplePrep()
ref = engine.resample_vals(x, u, order, sampler=s, path="int8_table", prep=prep)
torch.cuda.synchronize()

def timed(fn, n=4):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]

out = torch.empty_like(ref)
print(f"one call: {timed(lambda: engine.resample_vals(x, u, order, sampler=s, out=out, path='int8_table', prep=prep)):.2f} ms")
for nslab in (2, 3, 4):
    groups = -(-nrep // 128)
    per = -(-groups // nslab) * 128
    cuts = [(a, min(nrep, a + per)) for a in range(0, nrep, per)]
    streams = [torch.cuda.Stream() for _ in cuts]
    def run():
        cur = torch.cuda.current_stream()
        ev = torch.cuda.Event(); ev.record(cur)
        for st, (a, b) in zip(streams, cuts):
            st.wait_event(ev)
            with torch.cuda.stream(st):
                engine.resample_vals(x, u, order, sampler=s.rows(a, b), out=out[a:b], path="int8_table", prep=prep)
            e = torch.cuda.Event(); e.record(st); cur.wait_event(e)
    t = timed(run)
    torch.cuda.synchronize()
    print(f"{len(cuts)} slabs {cuts} on {len(cuts)} streams: {t:.2f} ms   same bits: {torch.equal(out, ref)}")
