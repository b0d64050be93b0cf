import sys, time, cProfile, pstats, io
sys.path.insert(0, ".")
import torch
import thermoextrap_amd as xtrap
from thermoextrap_amd import engine
from thermoextrap_amd.data import DeviceDataArray
from tools.bench_states import _state_xu
S, N, C, order, nrep = 64, 1_000_000, 4, 3, 100
sts = []
for s in range(S):
    xx, uu = _state_xu(torch, s, N, C)
    d = xtrap.DataCentralMomentsVals.from_vals(xv=DeviceDataArray(xx, ("rec", "val")), uv=DeviceDataArray(uu, ("rec",)), order=order, central=True)
    sts.append(xtrap.beta.factory_extrapmodel(1.0 + 0.1 * s, d))
coll = xtrap.models.StateCollection(sts)
def step(i):
    return xtrap.gpr_input.input_GP_from_states(coll, n_rep=nrep, sampler={"nrep": nrep, "device": True, "seed": 100 + i})
for i in range(3): step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20): step(10 + i)
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / 20 * 1e3)
pr = cProfile.Profile(); pr.enable()
for i in range(20): step(40 + i)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
