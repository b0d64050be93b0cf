#!/usr/bin/env python
"""bench.py -- samples/s of the central-comoment bootstrap hot path (BASELINE.json metric) on MI355X.

One "step" = `ExtrapModel.resample({"nrep": nrep}).derivs()` through the drop-in API on a state
point whose samples are already resident in HBM: draw the sampler (device multinomial: tile-count
kernels), run the bootstrap (txm_resample_vals: window pre-pass with the precision guard, Philox
stage 3 fused into the contraction, finalize), evaluate the derivative table on the replicate states
(txm_eval_poly) and copy the (order+1, nrep, N_obs) derivatives to the host.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config north|c2|c4|c3|c5] [--mode states|replicas]

Multi-GPU (one process per GPU, RCCL): `--gpus N` with N > 1 starts the N ranks itself when it is not
already running under torch.distributed.run (child process, before anything touches the GPU) and
relays rank 0's JSON line; under torchrun (WORLD_SIZE set) it is a rank.
  --mode states   (default) every rank bootstraps its own state point (own data, own nrep replicates);
                  the replicate-state slabs are all-gathered at the end of the step.  "scaling": "weak",
                  value = ranks * N_samp / time.
  --mode replicas every rank holds the same state point and bootstraps its contiguous nrep / ranks replicates of
                  the ONE sampler stream (thermoextrap_amd.distributed.sharded_bootstrap: same seed, replicate
                  offset per rank -- the gathered result is the one-GPU result bit for bit); one all-gather.
                  "scaling": "strong", value = N_samp / time.
"""

from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_PEAK_TFLOPS = 78.6  # MI355X FP64 vector = matrix peak (77.6 measured, tools/mfma_f64_peak3.hip)
INT8_PEAK_TOPS = 5000.0  # MI355X_MICROARCH.md: I8 MFMA = 2x the BF16 rate (~2.5 PF dense)

# BASELINE.json configs: shapes of the per-sample bootstrap legs
CONFIGS = {
    "north": dict(n_samp=1e8, n_obs=32, order=4, nrep=1000),   # the metric's own configuration
    "c2": dict(n_samp=1e7, n_obs=8, order=4, nrep=200),
    "c4": dict(n_samp=1e8, n_obs=32, order=6, nrep=1000),     # + per-replicate <dx/dq> (volume callback)
}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=5)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--config", default="north", choices=sorted(CONFIGS) + ["c3", "c5"])
    p.add_argument("--mode", default="states", choices=["states", "replicas"])
    p.add_argument("--n-samp", type=float, default=None)
    p.add_argument("--n-obs", type=int, default=None)
    p.add_argument("--order", type=int, default=None)
    p.add_argument("--nrep", type=int, default=None)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-fp64-leg", action="store_true", help="skip timing the same shape with the FP64 kernel")
    p.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU baseline duration")
    p.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0: min(host cores, N_obs) -- the reference threads over observables only)")
    p.add_argument("--dry-run", action="store_true",
                   help="no GPU: ranks rendezvous over gloo and run the sharded step with a stand-in compute (tests the launcher)")
    return p.parse_args()


def dry_run(args):
    """The multi-rank plumbing of this script without a GPU (tests/test_distributed_cpu.py): rendezvous (gloo),
    thermoextrap_amd.distributed.run_step with a stand-in compute, barrier + max-over-ranks timing, rank 0's line."""
    import torch
    import torch.distributed as dist

    from thermoextrap_amd import distributed as txd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("gloo")
    rank, world = txd.world()
    nrep = 8
    if args.config in ("c3", "c5"):
        # the state-sharded configurations (tools/bench_states.py): every rank owns a contiguous share of the S state
        # points, one all-gather of the per-state blocks ends a step -- the gather helpers of the product, stand-in blocks
        S = 16 if args.config == "c3" else 64
        mine = txd.shard_range(S, rank, world)
        counts = txd.all_gather_ints(len(mine))                 # what input_GP_from_states(sharded="local") does
        assert counts == txd.shard_counts(S, world) and sum(counts[:rank]) == mine.start
        block = torch.zeros((len(mine), 3), dtype=torch.float64)
        block[:, 0] = torch.arange(mine.start, mine.stop, dtype=torch.float64)
        block[:, 1] = float(rank)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            full = txd.all_gather_slabs(block, counts)
        dt = time.perf_counter() - t0
        assert full[:, 0].tolist() == list(range(S)), full[:, 0].tolist()
        if rank == 0:
            print(json.dumps({"metric": "dry-run", "config": args.config, "n_gpus": world, "steps": args.steps, "states": S,
                              "states_per_rank": counts, "ranks_seen": sorted({int(v) for v in full[:, 1].tolist()})}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    def compute(n, seed, rep0):
        out = torch.full((n, 2, 2, 3), float(rank), dtype=torch.float64)
        out[:, 1, 1, 1] = torch.arange(rep0, rep0 + n, dtype=torch.float64)  # the stream replicates of this slab
        return out

    t0 = time.perf_counter()
    for i in range(args.steps):
        out = txd.run_step(args.mode, compute, nrep, 100 + i)
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    want = world * nrep if args.mode == "states" else nrep
    assert out.shape[0] == want, (out.shape, want)
    # one stream for the whole job: the gathered slabs cover stream replicates 0 .. rows - 1 exactly once, in order
    assert out[:, 1, 1, 1].tolist() == list(range(want)), out[:, 1, 1, 1].tolist()
    if rank == 0:
        print(json.dumps({"metric": "dry-run", "n_gpus": world, "steps": args.steps, "mode": args.mode,
                          "rows": int(out.shape[0]), "ranks_seen": sorted({int(v) for v in out[:, 0, 0, 0].tolist()})}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def spawn_ranks(n: int) -> int:
    """Start `n` ranks of this script under torch.distributed.run as a CHILD process (this process has
    not touched the GPU and never execs) and relay the JSON line rank 0 prints."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = os.environ.copy()
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    return r.returncode if line is not None or r.returncode else 1


def init_ranks(torch):
    """(world, rank, local_rank, dist): one rank per GPU over RCCL when started under torch.distributed.run, else (1, 0, 0,
    None).  Shared by every configuration (tools/bench_states.py for c3 / c5)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # one rank per GPU over RCCL.  TXM_BENCH_BACKEND=gloo is a rehearsal switch for a box with fewer GPUs than
        # ranks (the ranks then share cards and the final gather goes through host memory): plumbing test only
        backend = os.environ.get("TXM_BENCH_BACKEND", "nccl")
        ngpu = torch.cuda.device_count()
        if backend == "nccl" and local >= ngpu:
            raise SystemExit(f"rank {rank}: local rank {local} but only {ngpu} GPU(s) visible")
        local = local % max(ngpu, 1)
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
        world = dist.get_world_size()  # the rank count the process group reports
    else:
        torch.cuda.set_device(0)
    return world, rank, local, dist


def ranks_identity(torch, dist, world):
    """Which GPU every rank ran on (evidence of the rank count under RCCL: one distinct device per rank)."""
    try:
        props = torch.cuda.get_device_properties(torch.cuda.current_device())
        ident = f"{getattr(props, 'uuid', '')}|{getattr(props, 'pci_bus_id', '')}|{torch.cuda.current_device()}"
        if world > 1:
            objs = [None] * world
            dist.all_gather_object(objs, ident)
            return objs
        return [ident]
    except Exception as exc:  # noqa: BLE001
        return [f"unavailable: {exc}"]


def make_data(N, C, seed, torch):
    """Synthetic state point: u ~ N(174.85, 5.31^2) (ideal-gas
    scale, 3 % spread), x_c = a_c + b_c u + eps, generated on the device."""
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    u = torch.empty(N, dtype=torch.float64, device="cuda")
    u.normal_(174.85, 5.31, generator=g)
    a = torch.randn(C, dtype=torch.float64, device="cuda", generator=g)
    b = 1e-3 + 5e-4 * torch.randn(C, dtype=torch.float64, device="cuda", generator=g)
    x = torch.empty((N, C), dtype=torch.float64, device="cuda")
    step = 1 << 22
    for i0 in range(0, N, step):
        i1 = min(N, i0 + step)
        blk = x[i0:i1]
        blk.normal_(0.0, 0.05, generator=g)
        blk.add_(a[None, :]).addcmul_(u[i0:i1, None], b[None, :])
    return x, u


def cpu_baseline(C, order, nrep_full, seconds, ncores):
    """Reference-algorithm CPU baseline: the oracle's C restatement of cmomy's
    resample_vals (per-sample Pebay push, threads over replicates x observables as
    cmomy's parallel=True does) on a bounded sample, scaled linearly in N*nrep."""
    import numpy as np

    from oracle import oracle as orc

    rng = np.random.default_rng(0)
    nrep = 32
    # calibrate on a small run, then size N for ~`seconds`
    def run(N):
        u = rng.normal(174.85, 5.31, N)
        x = 0.2 + 1e-3 * u[:, None] + rng.normal(0, 0.05, (N, C))
        freq = orc.indices_to_freq(rng.choice(N, (nrep, N)), N)
        t0 = time.perf_counter()
        orc.resample_vals(x, u, freq, order, nthreads=ncores)
        return time.perf_counter() - t0

    t_small = run(20000)
    N = int(min(8_000_000, max(20000, 20000 * seconds / max(t_small, 1e-3))))
    t = run(N)
    # samples/s at nrep_full replicates: work is linear in N * nrep
    value = N * (nrep / nrep_full) / t
    return {
        "value": value,
        "unit": "samples/s",
        "cores": ncores,
        "kind": "port",
        "sample": f"oracle/cmomy_oracle.c resample_vals (Pebay push per sample), N={N}, N_obs={C}, order={order}, "
                  f"nrep={nrep} in {t:.2f} s on {ncores} threads; scaled linearly in N*nrep to nrep={nrep_full}",
    }


def csrc_sha():
    """sha256 (16 hex digits) over the kernel sources: ties a committed PMC figure to the code it was measured on."""
    from thermoextrap_amd._build import csrc_sha as f

    return f()


def _traffic_files(shape):
    """(dict, file name) of every committed PMC summary of this workload shape measured on THIS checkout's kernel sources,
    newest first; and the reason when one exists for other sources only."""
    import glob

    sha = csrc_sha()
    stale = None
    hits = []
    for f in sorted(glob.glob(str(ROOT / "profiles" / "*_traffic.json")), reverse=True):
        try:
            d = json.loads(Path(f).read_text())
        except Exception:  # noqa: BLE001
            continue
        wl = d.get("workload", {})
        got = (wl.get("n_samp", 100_000_000), wl.get("n_obs", 32), wl.get("order", 4), wl.get("nrep", 1000))
        if tuple(int(v) for v in got) != tuple(shape):
            continue
        if d.get("csrc_sha") != sha:
            stale = stale or f"profiles/{Path(f).name} was measured on other kernel sources (csrc_sha {d.get('csrc_sha')} != {sha})"
            continue
        hits.append((d, f"profiles/{Path(f).name} (csrc_sha {sha})"))
    return hits, stale


def pmc_traffic(kernel_prefix, shape):
    """HBM bytes per launch of ONE kernel from the newest committed PMC summary of this workload (separate rocprofv3
    --pmc passes of this same command, tools/collect_profiles.py) and the file it came from; (None, reason)
    if no committed profile matches this run's shape AND the kernel sources it was measured on (`csrc_sha` in the
    summary against the sources of this checkout: a PMC figure for other code is not reported).  NOT measured in this run."""
    hits, stale = _traffic_files(shape)
    for d, src in hits:
        # several instantiations can share the prefix (e.g. the FP64 kernel's empty listed-mode launches): the
        # one that moved the most bytes is the kernel of this workload
        vals = [v.get("hbm_bytes_per_launch") for k, v in d.get("kernels", {}).items() if k.startswith(kernel_prefix)]
        vals = [h for h in vals if h is not None]
        if vals:
            return max(vals), src
    return None, stale


# the kernels ONE txm_resample_vals call on the int8 path launches, by name prefix (txm_resample.hip: resample_vals_impl):
# the contraction kernel(s) of the path, the guard's FP64 kernel in listed mode (template mode 1: empty on ordinary data),
# the finalize kernels, the info word; the count-table generator on the table path; the pre-pass when the caller's block
# is not reused.  Every one of them runs once per 32-column group except the generator, the pivot and the info kernel.
_CALL_KERNELS = {
    "int8_table": ("txm::count_table_kernel", "txm::resample_i8g_kernel", "txm::resample_i8gn_kernel", "txm::resample_i8t_kernel"),
    "int8_fused": ("txm::resample_i8t_kernel",),
}
_CALL_COMMON = ("txm::resample_finalize_i8_kernel", "txm::resample_finalize_y_kernel", "txm::i8_info_kernel")
_CALL_PREPASS = ("txm::pivot_kernel", "txm::i8_stats_kernel", "txm::i8_table_kernel", "txm::i8_list_kernel")
_ONCE_PER_CALL = ("txm::count_table_kernel", "txm::i8_info_kernel", "txm::pivot_kernel")


def call_traffic(kernels: dict, int8_kernel: str, n_obs: int, prepass: bool = False, has_y: bool = False):
    """HBM bytes ONE bootstrap call moves, from a traffic summary's per-kernel per-launch figures: the SUM over the
    kernels of the call (round-5 verdict: the bench line carried one launch's 132 GB where the call moves 370).
    -> (total, {kernel: bytes per call}).  ``has_y``: the call carries a second matrix (config 4) -- its last pass is the
    `<J0, JN, w, true>` instance of the contraction kernel; a `<J0, JN, w, false>` sibling in the same summary belongs to another
    leg of the profiled command (the parity check bootstraps without the second matrix) and is left out."""
    groups = -(-int(n_obs) // 32)
    names = _CALL_KERNELS[int8_kernel] + _CALL_COMMON + (_CALL_PREPASS if prepass else ())
    ys_siblings = {k[: k.rfind(",")] for k in kernels if has_y and k.rstrip(">").endswith(", true") and "resample_i8" in k}
    per = {}
    for k, v in kernels.items():
        b = v.get("hbm_bytes_per_launch")
        if b is None:
            continue
        if has_y and "resample_i8" in k and k.rstrip(">").endswith(", false") and k[: k.rfind(",")] in ys_siblings:
            continue
        listed = k.startswith("txm::resample_kernel<") and k.rstrip(">").split(",")[-2].strip() == "1"  # RS_LISTED
        if not (listed or any(k.startswith(n) for n in names)):
            continue
        per[k] = b * (1 if any(k.startswith(n) for n in _ONCE_PER_CALL) else groups)
    return sum(per.values()), per


def pmc_call_traffic(shape, int8_kernel, prepass=False, has_y=False):
    hits, stale = _traffic_files(shape)
    for d, src in hits:
        total, per = call_traffic(d.get("kernels", {}), int8_kernel, shape[1], prepass, has_y)
        # (wide states: txm_resample_i8g.hip; narrow states, C <= 16: txm_resample_i8gn.hip)
        main_kernel = ("txm::resample_i8gn_kernel" if shape[1] <= 16 else "txm::resample_i8g_kernel") if int8_kernel == "int8_table" else "txm::resample_i8t_kernel"
        if any(k.startswith(main_kernel) for k in per):   # (a summary of this shape that saw the call's contraction kernel)
            return total, per, src
    return None, None, stale


def pmc_narrow_call_traffic(cfg, workload, int8_kernel=None):
    """(total, {kernel: bytes}, source) of ONE bootstrap call of a narrow configuration ("c2" / "c5") from the newest committed
    profiles/*_pmc_narrow.json (tools/narrow_pmc.sh: FETCH_SIZE / WRITE_SIZE passes over every kernel of the call, summed as
    call_traffic does) measured on THIS checkout's kernel sources and on the same workload and int8 kernel; (None, None, reason)
    otherwise.  NOT measured in this run."""
    import glob

    sha = csrc_sha()
    stale = None
    for f in sorted(glob.glob(str(ROOT / "profiles" / "*_pmc_narrow.json")), reverse=True):
        try:
            d = json.loads(Path(f).read_text())
            c = d["configs"][cfg]
        except Exception:  # noqa: BLE001
            continue
        if any(int(c.get("workload", {}).get(k, -1)) != int(v) for k, v in workload.items()) or c.get("call_hbm_bytes") is None:
            continue
        if int8_kernel is not None and c.get("call_kernel") != int8_kernel:
            continue
        if d.get("csrc_sha") != sha:
            stale = stale or f"profiles/{Path(f).name} was measured on other kernel sources (csrc_sha {d.get('csrc_sha')} != {sha})"
            continue
        return c["call_hbm_bytes"], c.get("call_kernels"), f"profiles/{Path(f).name} [{cfg}] (csrc_sha {sha})"
    return None, None, stale


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    if args.dry_run:
        return dry_run(args)
    if args.config in ("c3", "c5"):
        from tools import bench_states  # multi-state configurations: batched reduce / bootstrap over states

        return bench_states.main(args)

    import torch

    world, rank, local, dist = init_ranks(torch)

    import thermoextrap_amd as xtrap
    from thermoextrap_amd import distributed as txd
    from thermoextrap_amd import engine
    from thermoextrap_amd.moments import DeviceDataArray

    xtrap.require_gpu(torch.cuda.current_device())

    cfg = dict(CONFIGS[args.config])
    for k, v in (("n_samp", args.n_samp), ("n_obs", args.n_obs), ("order", args.order), ("nrep", args.nrep)):
        if v is not None:
            cfg[k] = v
    N, C, order, nrep = int(cfg["n_samp"]), int(cfg["n_obs"]), int(cfg["order"]), int(cfg["nrep"])
    K = order + 1
    replicas = args.mode == "replicas" and world > 1
    # states mode: every rank its own state point; replicas mode: the same state point on every rank
    x, u = make_data(N, C, seed=1000 + (0 if replicas else rank), torch=torch)

    # ---- the state point through the public drop-in API -----------------------------
    xv = DeviceDataArray(x, ("rec", "val"))
    uv = DeviceDataArray(u, ("rec",))
    data = xtrap.DataCentralMomentsVals.from_vals(xv=xv, uv=uv, order=order, central=True)
    xem = xtrap.beta.factory_extrapmodel(5.6, data)
    state = data.dxduave.device_values  # (C, 2, K)
    pivot = torch.cat([state[0, 0, 1:2], state[:, 1, 0]]).contiguous()  # {<u>, <x_c>}
    results = {}

    # C4: the volume callback's per-replicate <dx/dq> is a second, order-0 bootstrap over the same sampler
    dxdq = None
    if args.config == "c4":
        dxdq, _ = make_data(N, C, seed=5000 + rank, torch=torch)

    # live per-call timing of the bootstrap entry point inside the timed region: HIP events recorded on
    # the stream the kernels are launched on (torch's current stream), around every engine.resample_vals call
    boot_events = []
    recording = {"on": False}
    _resample_vals = engine.resample_vals

    def resample_vals_timed(*a, **kw):
        if not recording["on"]:
            return _resample_vals(*a, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = _resample_vals(*a, **kw)
        e1.record()
        boot_events.append((e0, e1))
        return out

    engine.resample_vals = resample_vals_timed

    from thermoextrap_amd import moments as cm

    phase_events = []  # per timed step: (start, after sampler, after bootstrap, after derivs + D2H)

    def mark():
        if not recording["on"]:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def one_bootstrap(n_rep, seed, rep0):
        # one sampler object per step (the tile-count kernel runs here), shared by the moments and the callback's <dx/dq>;
        # replicates rep0 .. rep0 + n_rep of the stream of `seed`: the gathered job equals one rank's result bit for bit
        e0 = mark()
        smp = cm.factory_sampler({"nrep": n_rep, "device": True, "seed": seed, "rep0": rep0}, data=xv, dim="rec")
        e1 = mark()
        if dxdq is not None:
            # the volume callback's per-replicate <dx/dq>: the second sample matrix of the same call (txm_resample_opts.y)
            st, results["dxdq"] = engine.resample_vals(x, u, order, sampler=smp.device_sampler, y=dxdq,
                                                       prep=data._cache.setdefault("resample_prep", engine.ResamplePrep()))
            boot = xem.new_like(data=data.new_like(dxduave=cm.CentralMomentsData(st, mom_ndim=2, dims=("rep", "val", "xmom", "umom")),
                                                   rec_dim="rep"))
        else:
            boot = xem.resample(sampler=smp)
        e2 = mark()
        results["derivs"] = boot.derivs(norm=False)  # host labelled array (order+1, rep, val)
        e3 = mark()
        if e0 is not None:
            phase_events.append((e0, e1, e2, e3))
        return boot.data.dxduave.device_values

    def step(i):
        # thermoextrap_amd.distributed.run_step: the sharding (and the final all-gather over xGMI) that the
        # 2-rank gloo test tests/test_distributed_cpu.py drives with a stand-in `compute`
        results["slabs"] = txd.run_step("replicas" if replicas else "states", one_bootstrap, nrep, 12345 + 1000 * i)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    recording["on"] = True
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    dt_local = dt
    recording["on"] = False
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = 1e3 * dt / args.steps
    value = (1 if replicas else world) * N / (dt / args.steps)
    nrep_rank = len(txd.shard_range(nrep, rank, world)) if replicas else nrep
    main_calls = [a.elapsed_time(b) for a, b in boot_events]
    t_boot = sum(main_calls) / max(len(main_calls), 1)
    info = engine.resample_info()
    nph = max(len(phase_events), 1)
    ph = [sum(ev[i].elapsed_time(ev[i + 1]) for ev in phase_events) / nph for i in range(3)]
    step_breakdown = {
        "sampler_tile_counts": ph[0],
        "bootstrap_call": ph[1],   # txm_resample_vals through the API: (pre-pass unless reused) + contraction + finalize
        "prepass_reused": bool(info.get("prep_reused")),
        "int8_kernel": info.get("kernel"),
        "derivs_and_d2h": ph[2],
        "host_and_gather": max(ms_per_step - sum(ph), 0.0),
        "how": "HIP events on the launch stream inside the timed steps; host_and_gather = wall clock per step minus the events",
    }

    # ---- separate event timings of the other kernels of a step (same stream) -------
    def timed(fn, reps):
        evs = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        ts = [a.elapsed_time(b) for a, b in evs]
        return sum(ts) / len(ts)  # mean ms

    sampler = engine.DeviceSampler(seed=0, nrep=nrep_rank, ndat=N)
    out = torch.empty((nrep_rank, C, 2, K), dtype=torch.float64, device="cuda")
    t_samp = timed(lambda: sampler.draw(seed=777), 3)
    # the FIRST bootstrap of a data object also runs the int8 path's pre-pass (one more read of the samples: pivot, window
    # table, guard flags); the timed steps reuse the block the data object keeps -- `prepass_reused` -- so this is the
    # one-time cost per data object that `value` does not contain
    def cold():
        engine.resample_vals(x, u, order, sampler=sampler, out=out, prep=engine.ResamplePrep(), y=dxdq)
    cold()
    step_breakdown["cold_call_ms"] = timed(cold, 2)
    step_breakdown["cold_call_note"] = ("one txm_resample_vals call on a data object without a pre-pass block (pivot estimate + pre-pass + "
                                        "contraction + finalize); bootstrap_call above is the same call with the block reused")
    t_red = timed(lambda: engine.reduce_vals(x, u, order), 10)
    path = info["path"]
    t_fp64 = None
    parity = None
    if path == "int8" and not args.no_fp64_leg:
        with engine.forced_path("fp64"):
            engine.resample_vals(x, u, order, sampler=sampler, pivot=pivot, out=out)
            t_fp64 = timed(lambda: engine.resample_vals(x, u, order, sampler=sampler, pivot=pivot, out=out), 2)
        # parity of what was timed (outside the timed region): the default dispatch against the FP64 kernel on the same
        # sampler draw, every replicate and column, relative to each comoment's natural scale sigma_x^a sigma_u^b
        got = engine.resample_vals(x, u, order, sampler=sampler, pivot=pivot)
        sx, su = x[: 1 << 20].std(dim=0), u[: 1 << 20].std()
        sc = torch.empty((C, 2, K), dtype=torch.float64, device="cuda")
        for b in range(K):
            sc[:, 0, b] = su**b
            sc[:, 1, b] = sx * su**b
        parity = {"max_scaled_abs_diff_vs_fp64_kernel": float(((got - out).abs() / (out.abs() + sc[None])).max()),
                  "over": f"{nrep_rank} replicates x {C} columns x 2 x {K} comoments, same sampler draw",
                  "tolerance_in_tests": 1e-12}
        del got

    shape = (N, C, order, nrep_rank)
    alg_bytes = 8.0 * N * (C + 1)                         # samples read once
    alg_flops = 2.0 * N * nrep_rank * K * (C + 1)         # dense contraction F.M

    def fp64_block(ms, kernel, measured):
        tf = alg_flops / (ms * 1e-3) / 1e12
        tr, src = pmc_traffic("txm::resample_kernel", shape)
        return {
            "kernel": kernel, "bound": "mfma", "pipe": "fp64", "achieved": tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": tf / FP64_PEAK_TFLOPS, "traffic": tr, "traffic_source": src, "ms": ms, "measured": measured,
            "algorithmic_flops": alg_flops, "algorithmic_bytes": alg_bytes,
            "hbm_achieved_GBs": alg_bytes / (ms * 1e-3) / 1e9, "hbm_frac": alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "note": "flops = 2*N*nrep*K*(N_obs+1) executed on v_mfma_f64_16x16x4_f64 (dense F.M, SURVEY 8(d)); fp64 "
                    "peak 78.6 TF (vector = matrix); max attainable HBM fraction of this workload is ~1 %",
        }

    live = "HIP events around every txm_resample_vals call of the timed steps (window pre-pass + guard list + memset + contraction + finalize)"
    roofline_fp64 = None
    if path == "int8":
        # executed int8 operations (txm_resample_i8t.hip): per workgroup (8 waves x 64 replicates) and k-step of 32
        # samples, every (column quad, power) fragment is contracted for both replicate halves (2 MFMAs) and every u-row
        # fragment (4 monomials) for both halves.  C > 16: 8 quads per 32-column group, orders 5..7 in two passes over the
        # sampler stream; narrow states (C <= 16): 1, 2 or 4 quads whose waves split the powers, one pass (four quads:
        # orders 6, 7 in two)
        if C <= 16 and K >= 2:
            nq = 1 if C <= 4 else 2 if C <= 8 else 4
            passes = [4, K - 4] if (nq == 4 and K >= 7) else [K]
            what = f"narrow state, {nq} column quad(s) whose waves split the powers; "
        else:
            nq = 8
            passes = {1: [1], 2: [2], 3: [3], 4: [4], 5: [5], 6: [3, 3], 7: [4, 3], 8: [4, 4]}[K]
            what = ""
        n_mfma = sum(2 * nq * jn + 2 * -(-jn // 4) for jn in passes)
        kname = "txm::resample_i8t_kernel"
        kdesc = (what + "bootstrap contraction on the int8 matrix pipe: 51-bit fixed-point words sliced once per sample, "
                 "byte-transposed by the LDS transposing read (ds_read_b64_tr_b8: 8 digit slots per word, 7 used), "
                 "exact int32 accumulation, Philox stage 3 fused")
        ksteps = -(-nrep_rank // 64) * (-(-N // 1024) * 32) * -(-C // 32)   # replicate groups x k-steps x column groups
        if info.get("kernel") == "int8_table" and C <= 16:
            # txm_resample_i8gn.hip: workgroup = 128 replicates x the state's 1, 2 or 4 column quads; per k-step 4 replicate
            # quarters x (quads x powers + u-row fragments of four monomials) MFMAs; four quads take orders >= 4 in two passes
            passes = [4, K - 4] if (nq == 4 and K >= 5) else [K]
            n_mfma = sum(4 * (nq * jn + -(-jn // 4)) for jn in passes)
            kname = "txm::resample_i8gn_kernel"
            kdesc = (what + "bootstrap contraction on the int8 matrix pipe over a count table in HBM (txm::count_table_kernel, part of "
                     f"the timed call): 128 replicates per workgroup, {len(passes)} pass(es), operands by LDS-DMA, two k-steps per LDS round trip")
            ksteps = -(-nrep_rank // 128) * (-(-N // 1024) * 32)
        elif info.get("kernel") == "int8_table":
            # txm_resample_i8g.hip: workgroup = 128 replicates x 32 columns; per k-step 4 replicate quarters x 8 column quads
            # per row set, the u-row in the words' dead byte (no extra MFMAs); K power row sets (+ the second matrix's) over
            # the fewest passes of <= 3; the count table of the call generated once (txm::count_table_kernel, inside the call)
            rows = K + (1 if dxdq is not None else 0)
            n_mfma = 32 * rows
            kname = "txm::resample_i8g_kernel"
            kdesc = ("bootstrap contraction on the int8 matrix pipe over a count table in HBM (txm::count_table_kernel, part of the "
                     f"timed call): {rows} row sets in {-(-rows // 3)} passes of <= 3, 128 replicates per workgroup, u-row in the dead "
                     "eighth byte, operands by LDS-DMA")
            ksteps = -(-nrep_rank // 128) * (-(-N // 1024) * 32) * -(-C // 32)
        i8_ops = 2.0 * 32 * 32 * 32 * n_mfma * ksteps
        tops = i8_ops / (t_boot * 1e-3) / 1e12
        table_split = None
        if info.get("kernel") == "int8_table":
            # the call = count-table generator (Philox-bound: 87 vector instructions per 12 draws) + contraction passes (int8 pipe) + finalize: the generator alone,
            # timed live on the same stream outside the timed region (txm_sampler_count_table: the launch the call makes)
            import ctypes as _ct
            from thermoextrap_amd import _lib as _tl
            _L = _tl.load()
            _tb = torch.empty(_L.txm_sampler_count_table_bytes(N, nrep_rank), dtype=torch.uint8, device="cuda")
            def _gen():
                _tl.check(_L.txm_sampler_count_table(_ct.byref(sampler.spec), engine._ptr(sampler.counts), 0, nrep_rank,
                                                     engine._ptr(_tb), engine._stream()), "count_table")
            _gen()
            t_gen = timed(_gen, 3)
            del _tb
            t_con = max(t_boot - t_gen, 1e-6)
            table_split = {"count_table_kernel_ms": t_gen, "count_table_written_GBs": N * (-(-nrep_rank // 128) * 128) / (t_gen * 1e-3) / 1e9,
                           "contraction_and_finalize_ms": t_con,
                           "contraction_frac_of_int8_peak": i8_ops / (t_con * 1e-3) / 1e12 / INT8_PEAK_TOPS,
                           "how": "generator timed alone with HIP events after the timed region; contraction = the call minus it "
                                  "(finalize and memsets included); profiles/*_kernel_stats.csv has the rocprofv3 per-kernel durations"}
        # HBM bytes of the timed CALL: the sum over its kernels (generator + every contraction pass + finalize ...), not one launch
        tr, tr_kernels, src = pmc_call_traffic(shape, info.get("kernel") or "int8_fused", prepass=not info.get("prep_reused"),
                                               has_y=dxdq is not None)
        if tr is None and C <= 16:  # narrow states: the counters of tools/narrow_pmc.sh (config 2's shape)
            tr, tr_kernels, src2 = pmc_narrow_call_traffic("c2", {"states": 1, "n_samp": N, "n_obs": C, "order": order, "nrep": nrep_rank},
                                                           info.get("kernel") or "int8_fused")
            src = src2 or src
        roofline = {
            "kernel": f"{kname} ({kdesc}) + pre-pass (unless reused) and finalize kernels",
            "bound": "mfma-i8", "pipe": "int8",
            "achieved": tops, "peak": INT8_PEAK_TOPS, "unit": "TOP/s", "frac": tops / INT8_PEAK_TOPS,
            "traffic": tr, "traffic_source": src,
            "traffic_ratio": (tr / alg_bytes) if tr else None,
            "hbm_executed_GBs": (tr / (t_boot * 1e-3) / 1e9) if tr else None,
            "hbm_executed_frac": (tr / (t_boot * 1e-3) / 1e9 / HBM_PEAK_GBS) if tr else None,
            "kernels": tr_kernels,
            "traffic_note": "traffic = HBM bytes of ONE timed bootstrap call = the sum over the kernels it launches (count-table "
                            "generator, every contraction pass, finalize, info; the pre-pass when not reused) of the per-launch "
                            "FETCH_SIZE x2 + WRITE_SIZE figures of the committed rocprofv3 --pmc summary; kernels = the addends; "
                            "traffic_ratio = traffic / algorithmic_bytes",
            "ms": t_boot, "measured": live,
            "executed_int8_ops": i8_ops,
            "algorithmic_flops": alg_flops, "algorithmic_bytes": alg_bytes,
            "fp64_equiv_tflops": alg_flops / (t_boot * 1e-3) / 1e12,
            "hbm_achieved_GBs": alg_bytes / (t_boot * 1e-3) / 1e9,
            "hbm_frac": alg_bytes / (t_boot * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "guard_windows": info["windows"], "guard_windows_fp64": info["windows_fp64"],
            "table_call_split": table_split,
            "note": "achieved = EXECUTED v_mfma_i32_32x32x32_i8 operations per second against the dense int8 peak "
                    "(2x bf16 = 5 POP/s); fp64_equiv_tflops = the algorithmic FP64 flops 2*N*nrep*K*(N_obs+1) per "
                    "second -- a speed, not a fraction of any roof.  12.5 % of the executed MFMA columns are the dead "
                    "eighth digit slot of the 8-byte word (DESIGN.md 4.2b)",
        }
        if t_fp64 is not None:
            roofline_fp64 = fp64_block(t_fp64, "txm::resample_kernel (the same shape forced onto the FP64 MFMA kernel, "
                                       "Philox stage 3 fused)", "HIP events around 2 forced-FP64 txm_resample_vals calls after the timed region")
    else:
        roofline = fp64_block(t_boot, "txm::resample_kernel (FP64 MFMA bootstrap contraction, Philox stage 3 fused)", live)
    roofline["path"] = path
    tr, src = pmc_traffic("txm::reduce_rowmajor_kernel", shape)
    roofline_reduce = {
        "kernel": "txm::reduce_rowmajor_kernel (one-pass power-sum reduction, the HBM-bound leg of the path)",
        "bound": "hbm",
        "achieved": alg_bytes / (t_red * 1e-3) / 1e9,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": alg_bytes / (t_red * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "traffic": tr, "traffic_source": src,
        "algorithmic_bytes": alg_bytes,
        "ms": t_red,
        "samples_per_s": N / (t_red * 1e-3),
    }

    ranks_seen = ranks_identity(torch, dist, world)
    # what every rank ran (the first SCALE record can be held against DESIGN 6's prediction rank by rank): its replicate slab
    # of the stream, the kernel the dispatch rule gave it, its own bootstrap-call and step times
    mine = {"rank": rank, "rep0": (txd.shard_range(nrep, rank, world).start if replicas else rank * nrep), "nrep": nrep_rank,
            "kernel": info.get("kernel") or path, "bootstrap_call_ms": round(t_boot, 3), "sampler_ms": round(ph[0], 3),
            "step_ms_local": round(1e3 * dt_local / args.steps, 3)}
    per_rank = [mine]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    if rank == 0:
        par = f"replicate-slabs x{world} (nrep/{world} per GPU, same state point)" if replicas else f"state-points x{world}"
        rec = {
            "metric": f"samples/s for order-{order} comoment bootstrap (N_samp x N_obs x nrep resample_vals)",
            "value": value,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong" if replicas else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "arithmetic": ("f64 in / out; inside the bootstrap kernel 51-bit fixed point as seven int8 digits with exact int32 "
                           "accumulation, FP64 partial sums; windows failing the precision guard on the FP64 MFMA kernel"
                           if path == "int8" else "f64 throughout"),
            "data": "synthetic",
            "config": {
                "workload": f"{args.config}: central-comoment bootstrap, N_samp={N:.0e}, N_obs={C}, order={order}, nrep={nrep}, "
                            "exact multinomial device sampler" + (", + per-replicate <dx/dq> on the same draw" if dxdq is not None else ""),
                "n_samp": N, "n_obs": C, "order": order, "nrep": nrep,
                "parallelism": par,
            },
            "replicate_samples_per_s": value * nrep,
            "ranks_seen": ranks_seen,
            "per_rank": per_rank,
            "sampler_ms": t_samp,
            "roofline": roofline,
            "roofline_reduce": roofline_reduce,
        }
        if roofline_fp64 is not None:
            rec["roofline_fp64_path"] = roofline_fp64
        if parity is not None:
            rec["parity_check"] = parity
        rec["step_breakdown_ms"] = step_breakdown
        if world == 1 and not args.no_cpu_baseline:
            # the reference's parallelism is over observables only (cmomy's numba threads span broadcast dims, never the sample
            # axis: SURVEY 8(d)) -- effective cores = min(N_obs, host cores)
            nthr = args.cpu_threads or min(os.cpu_count() or 1, C)
            rec["cpu_baseline"] = cpu_baseline(C, order, nrep, args.cpu_seconds, nthr)
            rec["cpu_baseline"]["host_cores"] = os.cpu_count()   # `cores` = the threads the baseline ran on
        rec["host_cores"] = os.cpu_count()
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
