#!/usr/bin/env python
"""Count-table generator (txm_sampler_count_table): bit-for-bit against txm_sampler_freq on small shapes, then its time at
bench sizes.   python tools/count_table_check.py [time [nocheck]]"""
import ctypes as ct, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import thermoextrap_amd as txa
from thermoextrap_amd import engine, _lib

txa.require_gpu(0)
L = _lib.load()


def table(s, rep_begin, nreps):
    nb = L.txm_sampler_count_table_bytes(s.ndat, nreps)
    t = torch.empty(nb, dtype=torch.uint8, device="cuda")
    _lib.check(L.txm_sampler_count_table(ct.byref(s.spec), engine._ptr(s.counts), rep_begin, nreps, engine._ptr(t), engine._stream()), "count_table")
    return t


def expand(t, N, nreps):
    """table -> [ceil(nreps/128)*128][N] counts (numpy), undoing the operand order and the slid last tile"""
    ntiles = (N + 1023) // 1024
    G = (nreps + 127) // 128
    a = t.cpu().numpy().reshape(G, ntiles, 32, 4, 2, 32, 16)  # g t s q half n32 b
    a = a.transpose(0, 3, 5, 1, 2, 4, 6).reshape(G * 128, ntiles, 1024)  # rep, tile, sample-in-window
    out = np.zeros((G * 128, N), dtype=np.int64)
    for tt in range(ntiles):
        b0 = min(tt * 1024, N - 1024)
        out[:, b0:b0 + 1024] += a[:, tt]
    return out


for (N, nrep, rep0, rb, nr) in [(4096, 130, 0, 0, 130), (5000, 200, 7, 0, 200), (5000, 200, 7, 128, 72), (1024 * 37 + 5, 64, 0, 0, 64),
                                (300000, 300, 1000, 128, 172), (2048, 1, 0, 0, 1)]:
    s = engine.DeviceSampler(123, nrep, N, rep0=rep0)
    f = s.freq().cpu().numpy()
    e = expand(table(s, rb, nr), N, nr)
    live = min(nrep - rb, e.shape[0])
    ok = np.array_equal(e[:live], f[rb:rb + live]) and not e[live:].any()
    print(f"N={N} nrep={nrep} rep0={rep0} slab=[{rb},{rb+nr}): {'OK' if ok else 'MISMATCH'}", flush=True)
    assert ok or len(sys.argv) > 2  # (a third argument: timing builds whose tables are wrong by construction)

if len(sys.argv) > 1:
    for (N, nrep) in [(20_000_000, 1000), (100_000_000, 256), (100_000_000, 1000)]:
        s = engine.DeviceSampler(1, nrep, N)
        nb = L.txm_sampler_count_table_bytes(N, nrep)
        t = torch.empty(nb, dtype=torch.uint8, device="cuda")
        ts = []
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(L.txm_sampler_count_table(ct.byref(s.spec), engine._ptr(s.counts), 0, nrep, engine._ptr(t), engine._stream()), "count_table")
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print(f"count table N={N} nrep={nrep}: {nb/1e9:.1f} GB  {min(ts[1:]):.2f} ms  ({nb/min(ts[1:])/1e9:.2f} TB/s written)", flush=True)
        del t
