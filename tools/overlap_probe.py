"""Can the sampler of step k + 1 hide behind the bootstrap of step k?  (round-2 verdict: "sampler on a second stream")
Two samplers / count tables, the next one drawn on a side stream while the main stream contracts the current one.
python tools/overlap_probe.py [N] [nrep] [steps]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

import thermoextrap_amd as txa
from bench import make_data
from thermoextrap_amd import engine as eng

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
nrep = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
C, order = 32, 4
txa.require_gpu(0)
x, u = make_data(N, C, 0, torch)
prep = eng.ResamplePrep()
out = torch.empty((nrep, C, 2, order + 1), dtype=torch.float64, device="cuda")
smp = [eng.DeviceSampler(1, nrep, N), eng.DeviceSampler(2, nrep, N)]
eng.resample_vals(x, u, order, sampler=smp[0], out=out, prep=prep)
torch.cuda.synchronize()


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def serial():
    for k in range(steps):
        smp[0].draw(100 + k)
        eng.resample_vals(x, u, order, sampler=smp[0], out=out, prep=prep)


side = torch.cuda.Stream()


def overlapped():
    main = torch.cuda.current_stream()
    ready = [torch.cuda.Event(), torch.cuda.Event()]   # counts of sampler b drawn
    freed = [torch.cuda.Event(), torch.cuda.Event()]   # bootstrap that read sampler b finished
    with torch.cuda.stream(side):
        smp[0].draw(100)
        ready[0].record(side)
    for k in range(steps):
        b = k & 1
        if k + 1 < steps:
            with torch.cuda.stream(side):
                if k >= 1:
                    side.wait_event(freed[1 - b])
                smp[1 - b].draw(101 + k)
                ready[1 - b].record(side)
        main.wait_event(ready[b])
        eng.resample_vals(x, u, order, sampler=smp[b], out=out, prep=prep)
        freed[b].record(main)


t_s = timed(serial)
t_o = timed(overlapped)
t_s2 = timed(serial)
print(f"N={N} nrep={nrep}: serial {t_s:.2f} ms/step, sampler of the next step on a side stream {t_o:.2f} ms/step, serial again {t_s2:.2f}")
