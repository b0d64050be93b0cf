"""bench.py --config c3 | c5: the multi-state configurations of BASELINE.json.

c3  16 state points x N = 1e7 potential-energy series, x_is_u, order 4: DataCentralMoments.from_vals over the
    (state, rec) array -- ONE launch of the 1-D reduction over all states (HBM-bound).
c5  64 state points (N = 1e6 samples x 4 observables each), order 3, nrep = 100: gpr_input.input_GP_from_states --
    bootstrap of all states in one launch, derivatives in one evaluation, covariance over replicates in one
    launch -- next to the reference-style serial loop over states (StateCollection.resample(batched=False) +
    per-state input_GP_from_state).

Prints one JSON line in bench.py's format (samples/s = states x N_samp / step time)."""
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


def main(args):
    import torch

    import thermoextrap_amd as xtrap
    from thermoextrap_amd import engine
    from thermoextrap_amd.moments import DeviceDataArray

    torch.cuda.set_device(0)
    xtrap.require_gpu(0)
    g = torch.Generator(device="cuda").manual_seed(0)
    if args.config == "c3":
        S, N, order = 16, int(args.n_samp or 1e7), int(args.order or 4)
        u = torch.empty((S, N), dtype=torch.float64, device="cuda")
        for s in range(S):
            u[s].normal_(-500.0 - 30.0 * s, 8.0 + s, generator=g)
        uv = DeviceDataArray(u, ("state", "rec"))
        out = {}

        def step():
            d = xtrap.DataCentralMoments.from_vals(uv=uv, xv=None, order=order, x_is_u=True, dim="rec", central=True)
            out["v"] = d.values.values      # host copy of the (state, xmom, umom) states

        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        t_k = _timed_events(torch, lambda: engine.reduce_vals_1d(u, order + 1), 10)
        alg = 8.0 * S * N
        rec = {
            "metric": "samples/s for order-4 central moments of 16 state points (x_is_u reduction)",
            "value": S * N / dt, "unit": "samples/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"c3: {S} states x N_samp={N:.0e} potential-energy series, x_is_u, order {order}, one batched launch",
                       "states": S, "n_samp": N, "order": order},
            "roofline": {"kernel": "txm::reduce_colmajor_kernel (1-D pivot-shifted power sums, all states in one launch) + pivot + finalize",
                         "bound": "hbm", "achieved": alg / (t_k * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg / (t_k * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "ms": t_k, "algorithmic_bytes": alg,
                         "measured": "HIP events around 10 txm_reduce_vals_1d calls"},
        }
        print(json.dumps(rec), flush=True)
        return

    # ---- c5
    S, N, C = 64, int(args.n_samp or 1e6), int(args.n_obs or 4)
    order, nrep = int(args.order or 3), int(args.nrep or 100)
    sts = []
    for s in range(S):
        uu = torch.empty(N, dtype=torch.float64, device="cuda").normal_(170.0 + s, 4.0 + 0.05 * s, generator=g)
        xx = 0.01 * uu[:, None] + 0.3 * torch.randn(N, C, generator=g, dtype=torch.float64, device="cuda")
        d = xtrap.DataCentralMomentsVals.from_vals(xv=DeviceDataArray(xx, ("rec", "val")), uv=DeviceDataArray(uu, ("rec",)),
                                                   order=order, central=True)
        sts.append(xtrap.beta.factory_extrapmodel(1.0 + 0.1 * s, d))
    coll = xtrap.models.StateCollection(sts)
    keep = {}

    def step(i):
        keep["gp"] = xtrap.gpr_input.input_GP_from_states(coll, n_rep=nrep, sampler={"nrep": nrep, "device": True, "seed": 100 + i})

    def serial(i):
        keep["sp"] = [xtrap.gpr_input.input_GP_from_state(st, n_rep=nrep, sampler={"nrep": nrep, "device": True, "seed": 100 + i + k})
                      for k, st in enumerate(coll)]

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    serial(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(max(1, args.steps // 2)):
        serial(1 + i)
    torch.cuda.synchronize()
    dts = (time.perf_counter() - t0) / max(1, args.steps // 2)
    xs = [st.data.xv.tensor for st in coll]
    us = [st.data.uv.tensor for st in coll]
    smp = engine.DeviceSampler(1, S * nrep, N)
    t_k = _timed_events(torch, lambda: engine.resample_vals_batched(xs, us, order, nrep=nrep, sampler=smp), 5)
    K = order + 1
    flops = 2.0 * S * N * nrep * K * (C + 1)
    # the FP64 kernel pads the observables to one 16-column MFMA block and the replicates to 64 per workgroup
    import os
    pack = 1 if os.environ.get("TXM_PACK", "1").startswith("0") or not (2 <= K <= 6 and C <= 8) else (4 if C <= 4 and K >= 3 else 2)
    exec_flops = 2.0 * S * N * (-(-nrep // 128) * 128) * -(-K // pack) * 16
    rec = {
        "metric": "samples/s for 64-state GP input (order-3 derivatives + covariance over 100 bootstrap replicates)",
        "value": S * N / dt, "unit": "samples/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"c5: {S} states x N_samp={N:.0e} x N_obs={C}, order {order}, nrep={nrep}: input_GP_from_states "
                               "(batched bootstrap + derivatives + covariance over replicates)",
                   "states": S, "n_samp": N, "n_obs": C, "order": order, "nrep": nrep},
        "serial_loop_ms_per_step": 1e3 * dts, "speedup_vs_serial_loop": dts / dt,
        "roofline": {"kernel": "txm::resample_kernel, batched over states (FP64 MFMA contraction, Philox stage 3 fused) + pivot + finalize",
                     "bound": "mfma", "pipe": "fp64", "achieved": flops / (t_k * 1e-3) / 1e12, "peak": 78.6, "unit": "TFLOP/s",
                     "frac": flops / (t_k * 1e-3) / 1e12 / 78.6, "executed_tflops": exec_flops / (t_k * 1e-3) / 1e12,
                     "traffic": None, "ms": t_k, "algorithmic_flops": flops,
                     "measured": "HIP events around 5 txm_resample_vals_batched calls",
                     "note": "achieved = algorithmic flops 2*S*N*nrep*K*(N_obs+1); the kernel executes ceil(K / pack) MFMAs of 16 columns "
                             f"(pack = {pack} powers of du per column) x 128 replicates per state and k-step (executed_tflops)"},
    }
    print(json.dumps(rec), flush=True)
