"""bench.py --config c3 | c5: the multi-state configurations of BASELINE.json.

c3  16 state points x N = 1e7 potential-energy series, x_is_u, order 4: DataCentralMoments.from_vals over the
    (state, rec) array -- ONE launch of the 1-D reduction over all states (HBM-bound).
c5  64 state points (N = 1e6 samples x 4 observables each), order 3, nrep = 100: gpr_input.input_GP_from_states --
    bootstrap of all states in one launch, derivatives in one evaluation, covariance over replicates in one
    launch -- next to the reference-style serial loop over states (StateCollection.resample(batched=False) +
    per-state input_GP_from_state).

Under torch.distributed.run (bench.py --config c3|c5 --gpus N) the STATE POINTS shard over the ranks: rank r builds and
bootstraps only its contiguous share, one all-gather of the per-state blocks ends a step
(input_GP_from_states(sharded="local")).  Prints one JSON line in bench.py's format (samples/s = states x N_samp /
step time, max over ranks) and exits non-zero when two nccl ranks report the same device."""
from __future__ import annotations

import json
import time

HBM_PEAK_GBS = 8000.0


def _timed_events(torch, fn, reps):
    evs = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    ts = [a.elapsed_time(b) for a, b in evs]
    return sum(ts) / len(ts)


def _state_u(torch, s, N):
    """potential-energy series of state point s (c3): seeded per state, so a rank's share equals the one-GPU rows."""
    g = torch.Generator(device="cuda").manual_seed(7000 + s)
    return torch.empty(N, dtype=torch.float64, device="cuda").normal_(-500.0 - 30.0 * s, 8.0 + s, generator=g)


def _state_xu(torch, s, N, C):
    """(x, u) of state point s (c5), seeded per state."""
    g = torch.Generator(device="cuda").manual_seed(9000 + s)
    uu = torch.empty(N, dtype=torch.float64, device="cuda").normal_(170.0 + s, 4.0 + 0.05 * s, generator=g)
    xx = 0.01 * uu[:, None] + 0.3 * torch.randn(N, C, generator=g, dtype=torch.float64, device="cuda")
    return xx, uu


def _timed_steps(torch, dist, world, step, warmup, steps):
    """W untimed + exactly K timed steps between barrier + synchronize on both sides; the MAX over ranks."""
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt / steps


def main(args):
    import sys

    import torch

    import bench
    import thermoextrap_amd as xtrap
    from thermoextrap_amd import distributed as txd
    from thermoextrap_amd import engine
    from thermoextrap_amd.moments import DeviceDataArray

    # state points shard over the ranks (BASELINE configs 3 and 5 are quoted "sharded 8xMI355X"): rank r owns the
    # contiguous share shard_range(S, r) of the states, builds ONLY their data, and one all-gather of the per-state
    # result blocks ends a step (reference loops: models.py:635-641, gpr_active/active_utils.py:896-925)
    world, rank, local, dist = bench.init_ranks(torch)
    xtrap.require_gpu(torch.cuda.current_device())
    ranks_seen = bench.ranks_identity(torch, dist, world)
    if world > 1 and dist.get_backend() == "nccl" and len(set(ranks_seen)) != world:
        if rank == 0:
            print(f"bench_states: {world} nccl ranks on {len(set(ranks_seen))} distinct device(s): {ranks_seen}", file=sys.stderr)
        dist.destroy_process_group()
        sys.exit(3)
    par = f"state-points over {world} rank(s)"
    if args.config == "c3":
        S, N, order = 16, int(args.n_samp or 1e7), int(args.order or 4)
        if world > S:
            raise SystemExit(f"c3 has {S} state points: at most {S} ranks")
        mine = txd.shard_range(S, rank, world)
        counts = txd.shard_counts(S, world)
        u = torch.stack([_state_u(torch, s, N) for s in mine])
        uv = DeviceDataArray(u, ("state", "rec"))
        out = {}

        def step(i):
            d = xtrap.DataCentralMoments.from_vals(uv=uv, xv=None, order=order, x_is_u=True, dim="rec", central=True)
            if world > 1:  # the (state, xmom, umom) blocks of all ranks, on every rank
                out["v"] = txd.all_gather_slabs(d.dxduave.device_values, counts).cpu().numpy()
            else:
                out["v"] = d.values.values      # host copy of the (state, xmom, umom) states

        dt = _timed_steps(torch, dist, world, step, args.warmup, args.steps)
        assert out["v"].shape[0] == S, out["v"].shape
        t_k = _timed_events(torch, lambda: engine.reduce_vals_1d(u, order + 1), 10)
        alg = 8.0 * len(mine) * N
        if rank == 0:
            rec = {
                "metric": "samples/s for order-4 central moments of 16 state points (x_is_u reduction)",
                "value": S * N / dt, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": 1e3 * dt, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
                "data": "synthetic",
                "config": {"workload": f"c3: {S} states x N_samp={N:.0e} potential-energy series, x_is_u, order {order}, one batched launch per rank",
                           "states": S, "n_samp": N, "order": order, "parallelism": par, "states_per_rank": counts},
                "ranks_seen": ranks_seen,
                "roofline": {"kernel": "txm::reduce_colmajor_kernel (1-D pivot-shifted power sums, all states of the rank in one launch) + pivot + finalize",
                             "bound": "hbm", "achieved": alg / (t_k * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": alg / (t_k * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "ms": t_k, "algorithmic_bytes": alg,
                             "measured": "HIP events around 10 txm_reduce_vals_1d calls (rank 0's states)"},
            }
            print(json.dumps(rec), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- c5
    S, N, C = 64, int(args.n_samp or 1e6), int(args.n_obs or 4)
    order, nrep = int(args.order or 3), int(args.nrep or 100)
    if world > S:
        raise SystemExit(f"c5 has {S} state points: at most {S} ranks")
    mine = txd.shard_range(S, rank, world)
    counts = txd.shard_counts(S, world)
    sts = []
    for s in mine:
        xx, uu = _state_xu(torch, s, N, C)
        d = xtrap.DataCentralMomentsVals.from_vals(xv=DeviceDataArray(xx, ("rec", "val")), uv=DeviceDataArray(uu, ("rec",)),
                                                   order=order, central=True)
        sts.append(xtrap.beta.factory_extrapmodel(1.0 + 0.1 * s, d))
    coll = xtrap.models.StateCollection(sts)
    keep = {}

    def step(i):
        keep["gp"] = xtrap.gpr_input.input_GP_from_states(coll, n_rep=nrep, sampler={"nrep": nrep, "device": True, "seed": 100 + i},
                                                          sharded="local" if world > 1 else False)

    def serial(i):
        keep["sp"] = [xtrap.gpr_input.input_GP_from_state(st, n_rep=nrep, sampler={"nrep": nrep, "device": True, "seed": 100 + i + k})
                      for k, st in enumerate(coll)]

    dt = _timed_steps(torch, dist, world, step, args.warmup, args.steps)
    assert keep["gp"][1].shape == (S * (order + 1), C), keep["gp"][1].shape
    dts = None
    if world == 1:
        serial(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(max(1, args.steps // 2)):
            serial(1 + i)
        torch.cuda.synchronize()
        dts = (time.perf_counter() - t0) / max(1, args.steps // 2)
    xs = [st.data.xv.tensor for st in coll]
    us = [st.data.uv.tensor for st in coll]
    S_loc = len(coll)
    smp = engine.DeviceSampler(1, S_loc * nrep, N)
    kprep = engine.ResamplePrep()  # as in the steps: the collection keeps the int8 path's pre-pass block
    engine.resample_vals_batched(xs, us, order, nrep=nrep, sampler=smp, prep=kprep)  # fills the block
    t_k = _timed_events(torch, lambda: engine.resample_vals_batched(xs, us, order, nrep=nrep, sampler=smp, prep=kprep), 5)
    info = engine.batched_info()
    K = order + 1
    flops = 2.0 * S_loc * N * nrep * K * (C + 1)
    if rank == 0:
        rec = {
            "metric": "samples/s for 64-state GP input (order-3 derivatives + covariance over 100 bootstrap replicates)",
            "value": S * N / dt, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"c5: {S} states x N_samp={N:.0e} x N_obs={C}, order {order}, nrep={nrep}: input_GP_from_states "
                                   "(batched bootstrap + derivatives + covariance over replicates)",
                       "states": S, "n_samp": N, "n_obs": C, "order": order, "nrep": nrep, "parallelism": par,
                       "states_per_rank": counts},
            "ranks_seen": ranks_seen,
            "roofline": _c5_roofline(info, flops, t_k, S_loc, N, C, K, nrep),
        }
        if dts is not None:
            rec["serial_loop_ms_per_step"] = 1e3 * dts
            rec["speedup_vs_serial_loop"] = dts / dt
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.destroy_process_group()


def _c5_roofline(info, flops, t_k, S, N, C, K, nrep):
    import os

    if info.get("path") == "int8":
        # narrow states on the int8 kernel: NQ column quads whose waves split the powers, one pass; per workgroup (64
        # replicates) and 32-sample k-step 2 NQ K x-tile MFMAs + 2 ceil(K / 4) u-row MFMAs
        nq = 1 if C <= 4 else 2 if C <= 8 else 4
        n_mfma = 2 * nq * K + 2 * -(-K // 4)
        ops = 2.0 * 32 * 32 * 32 * n_mfma * S * -(-nrep // 64) * (-(-N // 1024) * 32)
        tops = ops / (t_k * 1e-3) / 1e12
        import bench
        tr, tr_kernels, src = bench.pmc_narrow_call_traffic("c5", {"states": S, "n_samp": N, "n_obs": C, "order": K - 1, "nrep": nrep}, "int8_fused")
        alg_bytes = 8.0 * S * N * (C + 1)
        return {"kernel": "txm::resample_i8t_kernel, narrow-state variant with the state on the grid (int8 MFMA, Philox stage 3 fused) "
                          "+ batched pre-pass and finalize",
                "bound": "mfma-i8", "pipe": "int8", "achieved": tops, "peak": 5000.0, "unit": "TOP/s", "frac": tops / 5000.0,
                "traffic": tr, "traffic_source": src, "traffic_ratio": (tr / alg_bytes) if tr else None, "kernels": tr_kernels,
                "algorithmic_bytes": alg_bytes, "hbm_executed_GBs": (tr / (t_k * 1e-3) / 1e9) if tr else None,
                "traffic_note": "HBM bytes of ONE batched bootstrap call (pre-pass block reused) = the sum over its kernels of the FETCH_SIZE x2 + "
                                "WRITE_SIZE figures of tools/narrow_pmc.sh's committed summary",
                "ms": t_k, "executed_int8_ops": ops, "algorithmic_flops": flops,
                "fp64_equiv_tflops": flops / (t_k * 1e-3) / 1e12,
                "guard_windows_fp64": info.get("windows_fp64"),
                "prepass_reused": info.get("prep_reused"),
                "measured": "HIP events around 5 txm_resample_vals_batched_opts calls (rank 0's states; pre-pass block kept by the caller as in the steps)"}
    pack = 1 if not (2 <= K <= 6 and C <= 8) else (4 if C <= 4 and K >= 3 else 2)
    # the FP64 kernel pads the observables to one 16-column MFMA block and the replicates to 64 per workgroup
    exec_flops = 2.0 * S * N * (-(-nrep // 128) * 128) * -(-K // pack) * 16
    return {"kernel": "txm::resample_kernel, batched over states (FP64 MFMA contraction, Philox stage 3 fused) + pivot + finalize",
            "bound": "mfma", "pipe": "fp64", "achieved": flops / (t_k * 1e-3) / 1e12, "peak": 78.6, "unit": "TFLOP/s",
            "frac": flops / (t_k * 1e-3) / 1e12 / 78.6, "executed_tflops": exec_flops / (t_k * 1e-3) / 1e12,
            "traffic": None, "ms": t_k, "algorithmic_flops": flops,
            "measured": "HIP events around 5 txm_resample_vals_batched calls (rank 0's states)",
            "note": "achieved = algorithmic flops 2*S*N*nrep*K*(N_obs+1); the kernel executes ceil(K / pack) MFMAs of 16 columns "
                    f"(pack = {pack} powers of du per column) x 128 replicates per state and k-step (executed_tflops)"}
