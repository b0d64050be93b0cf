"""Time the tile-count sampler alone:  python tools/sampler_time.py [N] [nrep] [reps]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from thermoextrap_amd import engine

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
nrep = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
engine.DeviceSampler(1, nrep, N)
torch.cuda.synchronize()
ts = []
for i in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    s = engine.DeviceSampler(100 + i, nrep, N)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
c = s.counts if hasattr(s, "counts") else None
print(f"N={N:.3g} nrep={nrep}: tile counts in {min(ts):.3f} ms (min of {reps}), mean {sum(ts)/len(ts):.3f} ms")
