"""Where the host time of a config-5 step goes (GPU box): python tools/c5_host_profile.py"""
import cProfile
import pstats
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

import thermoextrap_amd as xtrap
from thermoextrap_amd.moments import DeviceDataArray

xtrap.require_gpu(0)
S, N, C, order, nrep = 64, 1_000_000, 4, 3, 100
sts = []
for s in range(S):
    g = torch.Generator(device="cuda").manual_seed(s)
    u = 174.85 + 5.31 * torch.randn(N, generator=g, dtype=torch.float64, device="cuda")
    x = 0.2 + 1e-3 * u[:, None] + 0.05 * torch.randn(N, C, generator=g, dtype=torch.float64, device="cuda")
    d = xtrap.DataCentralMomentsVals.from_vals(xv=DeviceDataArray(x, ("rec", "val")), uv=DeviceDataArray(u, ("rec",)), order=order, central=True)
    sts.append(xtrap.beta.factory_extrapmodel(1.0 + 0.1 * s, d))
coll = xtrap.models.StateCollection(sts)
for i in range(3):
    xtrap.gpr_input.input_GP_from_states(coll, n_rep=nrep, sampler={"nrep": nrep, "device": True, "seed": i})
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(10):
    xtrap.gpr_input.input_GP_from_states(coll, n_rep=nrep, sampler={"nrep": nrep, "device": True, "seed": 10 + i})
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
