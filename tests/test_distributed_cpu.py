"""world_size-2 gloo tests (CPU) of the multi-GPU sharding helpers: the N>1 path of
bench.py / thermoextrap_amd.distributed is exercised with synthetic slabs, since the
compute itself needs a GPU."""

import os
import subprocess
import sys
import textwrap
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent

WORKER = textwrap.dedent(
    """
    import os, sys
    sys.path.insert(0, os.environ["TXM_ROOT"])
    import torch, torch.distributed as dist
    from thermoextrap_amd import distributed as D
    dist.init_process_group("gloo")
    rank, w = D.world()
    assert w == 2
    # replicate-sharded bootstrap: nrep = 7 -> shares (4, 3); slab content encodes (rank, local index, stream replicate)
    assert D.replicate_offsets(7, w) == [0, 4]
    def compute(n, seed, rep0):
        assert seed == 123 and rep0 == (0, 4)[rank]          # ONE stream, contiguous replicate ranges
        out = torch.zeros((n, 3, 2, 5), dtype=torch.float64)
        out += rank * 100
        out[:, 0, 0, 0] += torch.arange(n, dtype=torch.float64)
        out[:, 2, 1, 4] = torch.arange(rep0, rep0 + n, dtype=torch.float64)
        return out
    full = D.sharded_bootstrap(compute, nrep=7, seed=123)
    assert full.shape == (7, 3, 2, 5)
    want = torch.tensor([0, 1, 2, 3, 100, 101, 102], dtype=torch.float64)
    assert torch.equal(full[:, 0, 0, 0], want), full[:, 0, 0, 0]
    assert full[:, 2, 1, 4].tolist() == [0, 1, 2, 3, 4, 5, 6]    # stream replicates 0..6, each exactly once, in order
    assert torch.equal(full[:4, 1], torch.zeros(4, 2, 5, dtype=torch.float64))
    assert torch.equal(full[4:, 1], torch.full((3, 2, 5), 100.0, dtype=torch.float64))
    # the step function bench.py runs, both modes (stand-in compute)
    def compute2(n, seed, rep0):
        out = torch.full((n, 2, 2, 3), float(seed % 1000), dtype=torch.float64)
        out[:, 0, 0, 0] = torch.arange(n, dtype=torch.float64) + 1000 * rank
        out[:, 1, 0, 2] = torch.arange(rep0, rep0 + n, dtype=torch.float64)
        return out
    st = D.run_step("states", compute2, 5, 40)
    assert st.shape == (10, 2, 2, 3)
    assert st[:, 0, 0, 0].tolist() == [0, 1, 2, 3, 4, 1000, 1001, 1002, 1003, 1004]
    assert st[:, 1, 1, 1].eq(40.0).all()                          # one seed for the whole job
    assert st[:, 1, 0, 2].tolist() == list(range(10))             # state r owns stream replicates r * nrep ...
    rp = D.run_step("replicas", compute2, 5, 40)
    assert rp.shape == (5, 2, 2, 3) and rp[:, 0, 0, 0].tolist() == [0, 1, 2, 1000, 1001]
    assert rp[:, 1, 1, 1].eq(40.0).all() and rp[:, 1, 0, 2].tolist() == [0, 1, 2, 3, 4]
    assert D.broadcast_int(77 + rank) == 77
    # StateCollection.resample(sharded=True): the method itself, with a stand-in for the GPU bootstrap of a rank's
    # sub-collection (the real one is held against the unsharded call on the GPU in tests/test_batched_gpu.py)
    import numpy as np
    from thermoextrap_amd import models as M
    class FakeDX:
        def __init__(self, t): self.device_values = t; self.dims = ("rep", "val", "xmom", "umom")
    class FakeData:
        rec_dim = "rep"; meta = None
        def __init__(self, dx=None): self.dxduave = dx
        def new_like(self, **kw): return FakeData(kw.get("dxduave"))
    class FakeState:
        order = 1; alpha0 = 0.0; derivatives = None; minus_log = False; alpha_name = "beta"
        def __init__(self, i, data=None): self.i = i; self.data = data or FakeData()
        def new_like(self, **kw): return FakeState(self.i, kw.get("data"))
    class FakeColl(M.StateCollection):
        calls = []
        def _batch_eligible(self): return None
        def resample(self, sampler, batched=None, **kws):
            if kws.get("sharded"):
                return super().resample(sampler, batched=batched, **kws)
            # what _resample_batched / the serial loop do with (spec, state0): state s draws replicates (state0 + s) * nrep ...
            state0, nrep = kws.pop("state0", 0), sampler["nrep"]
            FakeColl.calls.append((sampler["seed"], sampler.get("device"), state0, len(self)))
            sts = []
            for s, st in enumerate(self.states):
                t = torch.zeros((nrep, 2, 2, 2), dtype=torch.float64)
                t[:, 0, 0, 0] = torch.arange(nrep, dtype=torch.float64) + (state0 + s) * nrep
                t[:, 1, 1, 1] = float(sampler["seed"])
                sts.append(FakeState(st.i, FakeData(FakeDX(t))))
            return FakeColl(states=tuple(sts), kws=self.kws)
    import thermoextrap_amd.moments as cm
    cm.CentralMomentsData = lambda t, mom_ndim, dims: FakeDX(t)
    coll = FakeColl(states=tuple(FakeState(i) for i in range(5)))
    out = coll.resample({"nrep": 3, "seed": 999}, sharded=True)
    assert FakeColl.calls == [(999, True, (0, 3)[rank], (3, 2)[rank])], FakeColl.calls
    got = torch.stack([st.data.dxduave.device_values[:, 0, 0, 0] for st in out.states]).reshape(-1)
    assert got.tolist() == list(range(15))                         # == the unsharded replicate ranges, in order
    FakeColl.calls.clear()
    out2 = coll.resample({"nrep": 3}, sharded=True)                # no seed: rank 0 draws one, every rank uses it
    seeds = {float(st.data.dxduave.device_values[0, 1, 1, 1]) for st in out2.states}
    assert len(seeds) == 1
    for bad in ({"nrep": 3, "device": False}, np.zeros((3, 4), dtype=np.int64)):
        try:
            coll.resample(bad, sharded=True)
            raise SystemExit("sharded resample accepted a sampler it cannot split consistently")
        except ValueError:
            pass
    # state sharding: 5 states over 2 ranks, results come back in order on every rank
    states = list(range(5))
    outs = D.sharded_states(states, lambda s: torch.full((2, 2), float(s * s)))
    assert [float(o[0, 0]) for o in outs] == [0.0, 1.0, 4.0, 9.0, 16.0]
    # uneven slabs without counts
    g = D.all_gather_slabs(torch.full((rank + 1, 2), float(rank)))
    assert g.shape == (3, 2) and g[:, 0].tolist() == [0.0, 1.0, 1.0]
    # sharded_reduce: rank 0's pivot is the one every rank uses, the per-rank sums are gathered in rank order, every rank gets
    # the same state -- here with float64 stand-ins for the three device calls (the kernels themselves: tests/test_push_shard_gpu.py)
    import math
    g = torch.Generator().manual_seed(17)
    xx = 3.0 + torch.randn(1001, 2, generator=g, dtype=torch.float64); uu = 5.0 + 2.0 * torch.randn(1001, generator=g, dtype=torch.float64)
    sh = D.shard_range(1001, rank, w)
    Kk = 3
    def piv_fn(x_, u_): return torch.cat([u_[:7].mean()[None], x_[:7].mean(0)])
    def sums_fn(x_, u_, o_, p_, w_):
        du = u_ - p_[0]; dx = x_ - p_[1:]
        return torch.stack([torch.stack([torch.stack([(du ** j).sum() for j in range(o_ + 1)]),
                                         torch.stack([(dx[:, c] * du ** j).sum() for j in range(o_ + 1)])]) for c in range(x_.shape[1])])
    seen = {}
    def fin_fn(stack, p_):
        seen["stack"], seen["piv"] = stack.clone(), p_.clone()
        S = stack[0].clone()
        for i in range(1, stack.shape[0]): S = S + stack[i]
        return S
    tot = D.sharded_reduce(xx[sh.start:sh.stop], uu[sh.start:sh.stop], Kk - 1, ops=(piv_fn, sums_fn, fin_fn))
    assert torch.equal(seen["piv"], piv_fn(xx[:501], uu[:501]))                       # rank 0's shard decided the pivot
    assert seen["stack"].shape == (2, 2, 2, Kk)
    want = sums_fn(xx, uu, Kk - 1, seen["piv"], None)
    assert torch.allclose(tot, want, rtol=1e-12, atol=1e-9)
    ref_t = tot.clone(); dist.broadcast(ref_t, src=0); assert torch.equal(tot, ref_t)  # identical on every rank
    # an EMPTY shard (round-5 advice: a rank without samples failed locally and left the others in the broadcast): rank 0 holds
    # nothing, rank 1 everything -- the first non-empty rank estimates the pivot, the empty one adds zero sums, same state everywhere
    mine = slice(0, 0) if rank == 0 else slice(0, 1001)
    tot2 = D.sharded_reduce(xx[mine], uu[mine], Kk - 1, ops=(piv_fn, sums_fn, fin_fn))
    assert torch.equal(seen["piv"], piv_fn(xx, uu)) and torch.equal(seen["stack"][0], torch.zeros(2, 2, Kk, dtype=torch.float64))
    assert torch.allclose(tot2, sums_fn(xx, uu, Kk - 1, seen["piv"], None), rtol=1e-12, atol=1e-9)
    ref_t = tot2.clone(); dist.broadcast(ref_t, src=0); assert torch.equal(tot2, ref_t)
    try:
        D.sharded_reduce(xx[:0], uu[:0], Kk - 1, ops=(piv_fn, sums_fn, fin_fn))
        raise SystemExit("sharded_reduce accepted all-empty shards")
    except ValueError as e:
        assert "empty" in str(e)
    # input_GP_from_states(sharded=...): every raise is decided from gathered words, on EVERY rank -- a world larger than the
    # number of states, an empty local list or ineligible states on one rank must fail everywhere, not leave the other ranks
    # blocked in the all-gather (round-4 advice).  (No compute is reached: the checks come first.)
    from thermoextrap_amd import gpr_input as G
    plain = M.StateCollection(states=(FakeState(0),))               # one (ineligible) state for two ranks
    for kw in ({"sharded": True}, {"sharded": "local"}):
        c = plain if kw["sharded"] is True else M.StateCollection(states=(FakeState(0),) if rank == 0 else ())
        try:
            G.input_GP_from_states(c, sampler={"nrep": 4, "seed": 5}, **kw)
            raise SystemExit("sharded GP input accepted more ranks than states")
        except ValueError as e:
            assert "more ranks than states" in str(e), e
    two = M.StateCollection(states=(FakeState(0), FakeState(1)))    # a state per rank, none eligible: both ranks refuse
    try:
        G.input_GP_from_states(two, sampler={"nrep": 4, "seed": 5}, sharded=True)
        raise SystemExit("sharded GP input accepted ineligible states")
    except ValueError as e:
        assert "ranks without such states: [0, 1]" in str(e), e
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(os.environ["TXM_OUT"], f"rank{rank}.ok"), "w").write("ok")
    """
)


def test_shard_range_partition():
    from thermoextrap_amd import distributed as D

    for n in (0, 1, 7, 16, 1000):
        for w in (1, 2, 3, 8):
            parts = [D.shard_range(n, r, w) for r in range(w)]
            assert [i for p in parts for i in p] == list(range(n))
            sizes = [len(p) for p in parts]
            assert max(sizes) - min(sizes) <= 1
            assert D.shard_counts(n, w) == sizes
    with pytest.raises(ValueError):
        D.shard_range(4, 2, 2)


def test_two_rank_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, TXM_ROOT=str(ROOT), TXM_OUT=str(tmp_path), MASTER_ADDR="127.0.0.1")
    import socket

    with socket.socket() as sk:  # a free port (a fixed one collides with a concurrent run of this suite)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert (tmp_path / "rank0.ok").exists() and (tmp_path / "rank1.ok").exists()


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` (no WORLD_SIZE in the environment: what the driver runs) must start two ranks
    itself, as a child process, and relay rank 0's single JSON line.  --dry-run swaps the GPU work for a stand-in
    so that the launcher, the rendezvous and run_step are exercised here on the CPU."""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    for mode, rows in (("states", 16), ("replicas", 8)):
        r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--dry-run", "--mode", mode],
                           env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, lines
        rec = json.loads(lines[0])
        assert rec["n_gpus"] == 2 and rec["rows"] == rows and rec["ranks_seen"] == [0, 1] and rec["mode"] == mode


def test_bench_state_sharded_configs_start_their_own_ranks():
    """`bench.py --config c3|c5 --gpus 2` (BASELINE configs 3 and 5 are quoted sharded over 8 GPUs): the STATE POINTS go over
    the ranks.  --dry-run: launcher, rendezvous and the product's gather helpers on the CPU with stand-in blocks -- the
    all-gather returns all 16 / 64 states, in order, and rank 0 prints one line naming both ranks."""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    for cfg, S in (("c3", 16), ("c5", 64)):
        r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--config", cfg, "--gpus", "2", "--steps", "2", "--dry-run"],
                           env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, lines
        rec = json.loads(lines[0])
        assert rec["n_gpus"] == 2 and rec["states"] == S and rec["states_per_rank"] == [S // 2, S // 2]
        assert rec["ranks_seen"] == [0, 1] and rec["config"] == cfg
