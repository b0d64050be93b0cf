"""Worker of tests/test_sharded_gpu.py: one rank of a 2-rank gloo group, every rank on cuda:0.
Checks, on a real device, that what the ranks compute together equals the one-process result BIT FOR BIT:
  * StateCollection.resample(spec, sharded=True)  (5 states: shards of 3 and 2 -- and of 1 for a 5-rank layout emulated
    by hand below)                                  vs  StateCollection.resample(spec)
  * gpr_input.input_GP_from_states(sharded=True / "local")  vs  the one-process call (x, y and the block-diagonal noise)
  * distributed.run_step("replicas", ...)           vs  the full-nrep call, on both bootstrap kernels
  * distributed.sharded_reduce(x[shard], u[shard])  same bits on both ranks, the one-rank reduce of all samples to 1e-12
Writes <outdir>/rank<r>.json."""
import json
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import torch
import torch.distributed as dist


def main(outdir):
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    import thermoextrap_amd as xtrap
    from thermoextrap_amd import distributed as D, engine as eng
    from thermoextrap_amd.moments import DeviceDataArray

    xtrap.require_gpu(0)
    res = {"rank": rank, "world": world}

    def data(N, C, seed):
        g = torch.Generator(device="cuda").manual_seed(seed)
        u = 174.85 + 5.31 * torch.randn(N, generator=g, dtype=torch.float64, device="cuda")
        x = 0.2 + 1e-3 * u[:, None] + 0.05 * torch.randn(N, C, generator=g, dtype=torch.float64, device="cuda")
        return x, u

    # ---- state shards
    S, N, C, order, nrep = 5, 30_000, 3, 3, 20
    sts = []
    for s in range(S):
        x, u = data(N, C, 700 + s)
        d = xtrap.DataCentralMomentsVals.from_vals(xv=DeviceDataArray(x, ("rec", "val")), uv=DeviceDataArray(u, ("rec",)),
                                                   order=order, central=True)
        sts.append(xtrap.beta.factory_extrapmodel(1.0 + 0.25 * s, d))
    coll = xtrap.models.StateCollection(sts)
    spec = {"nrep": nrep, "seed": 99, "device": True}
    sharded = coll.resample(spec, sharded=True)
    whole = coll.resample(spec)
    res["states_equal"] = all(torch.equal(a.data.dxduave.device_values, b.data.dxduave.device_values)
                              for a, b in zip(sharded.states, whole.states))
    res["states_batch_equal"] = bool(torch.equal(sharded._batch["big"], whole._batch["big"]))
    # a spec without a seed: rank 0 draws one and broadcasts it -- every rank must end up with the same collection
    unseeded = coll.resample({"nrep": nrep, "device": True}, sharded=True)
    mine = unseeded._batch["big"].cpu()
    ref = mine.clone()
    dist.broadcast(ref, src=0)
    res["unseeded_consistent"] = bool(torch.equal(mine, ref))

    # ---- GP input of the collection (BASELINE config 5's call), states over the ranks: the whole collection on every
    # rank (sharded=True) and only the rank's own states (sharded="local") vs one process
    gspec = {"nrep": nrep, "seed": 1234, "device": True}
    one = xtrap.gpr_input.input_GP_from_states(coll, n_rep=nrep, sampler=gspec)
    share = D.shard_range(S, rank, world)
    local = xtrap.models.StateCollection(sts[share.start:share.stop])
    for key, got in (("gp_sharded", xtrap.gpr_input.input_GP_from_states(coll, n_rep=nrep, sampler=gspec, sharded=True)),
                     ("gp_local", xtrap.gpr_input.input_GP_from_states(local, n_rep=nrep, sampler=gspec, sharded="local")),
                     ("gp_log", xtrap.gpr_input.input_GP_from_states(local, n_rep=nrep, sampler=gspec, sharded="local", log_scale=True))):
        ref = one if key != "gp_log" else xtrap.gpr_input.input_GP_from_states(coll, n_rep=nrep, sampler=gspec, log_scale=True)
        res[key + "_equal"] = all(a.shape == b.shape and bool((a == b).all()) for a, b in zip(got, ref))

    # ---- replicate slabs of one state point, both kernels
    N2, C2, order2, nrep2 = 290_000, 32, 4, 130
    x, u = data(N2, C2, 5)
    for path in ("int8", "fp64"):
        with eng.forced_path(path):
            def compute(n, seed, rep0):
                return eng.resample_vals(x, u, order2, sampler=eng.DeviceSampler(seed, n, N2, rep0=rep0))
            got = D.run_step("replicas", compute, nrep2, 4321)
            full = compute(nrep2, 4321, 0)
        res[f"replicas_equal_{path}"] = bool(torch.equal(got, full))
    # ---- SAMPLE-sharded reduce (SURVEY 8(e) partition (4)): every rank reduces its half of one long series; the result is the
    # same on both ranks (bit for bit) and the one-rank state of all samples to 1e-12 of each comoment's scale
    N3, C3, order3 = 400_001, 6, 4
    x, u = data(N3, C3, 9)
    w = 0.5 + torch.rand(N3, dtype=torch.float64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    sh = D.shard_range(N3, rank, world)
    for key, ww in (("shred", None), ("shred_w", w)):
        got = D.sharded_reduce(x[sh.start:sh.stop], u[sh.start:sh.stop], order3, w=None if ww is None else ww[sh.start:sh.stop])
        full = eng.reduce_vals(x, u, order3, w=ww)
        other = got.cpu().clone()
        dist.broadcast(other, src=0)
        res[key + "_same_on_all_ranks"] = bool(torch.equal(got.cpu(), other))
        su = float(u.std())
        sx = x.std(dim=0)
        scale = torch.stack([torch.stack([sx[c] ** a * su ** b for b in range(order3 + 1)]) for a in range(2) for c in range(C3)]
                            ).reshape(2, C3, order3 + 1).permute(1, 0, 2)
        err = ((got - full).abs() / (full.abs() + scale)).max().item()
        res[key + "_max_err"] = err
    Path(outdir, f"rank{rank}.json").write_text(json.dumps(res))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
