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
    # replicate-sharded bootstrap: nrep = 7 -> shares (4, 3); slab content encodes (rank, seed, local index)
    seeds = D.replicate_seeds(123, w)
    assert len(set(seeds)) == w
    def compute(n, seed):
        assert seed == seeds[rank]
        out = torch.zeros((n, 3, 2, 5), dtype=torch.float64)
        out += rank * 100
        out[:, 0, 0, 0] += torch.arange(n, dtype=torch.float64)
        return out
    full = D.sharded_bootstrap(compute, nrep=7, seed=123)
    assert full.shape == (7, 3, 2, 5)
    want = torch.tensor([0, 1, 2, 3, 100, 101, 102], dtype=torch.float64)
    assert torch.equal(full[:, 0, 0, 0], want), full[:, 0, 0, 0]
    assert torch.equal(full[:4, 1], torch.zeros(4, 2, 5, dtype=torch.float64))
    assert torch.equal(full[4:, 1], torch.full((3, 2, 5), 100.0, dtype=torch.float64))
    # the step function bench.py runs, both modes (stand-in compute)
    def compute2(n, seed):
        out = torch.full((n, 2, 2, 3), float(seed % 1000), dtype=torch.float64)
        out[:, 0, 0, 0] = torch.arange(n, dtype=torch.float64) + 1000 * rank
        return out
    st = D.run_step("states", compute2, 5, 40)
    assert st.shape == (10, 2, 2, 3)
    assert st[:, 0, 0, 0].tolist() == [0, 1, 2, 3, 4, 1000, 1001, 1002, 1003, 1004]
    assert st[:5, 1, 1, 1].eq(40.0).all() and st[5:, 1, 1, 1].eq(41.0).all()      # seed + rank
    rp = D.run_step("replicas", compute2, 5, 40)
    assert rp.shape == (5, 2, 2, 3) and rp[:, 0, 0, 0].tolist() == [0, 1, 2, 1000, 1001]
    sd = D.replicate_seeds(40, w)
    assert rp[0, 1, 1, 1] == float(sd[0] % 1000) and rp[4, 1, 1, 1] == float(sd[1] % 1000)
    # state sharding: 5 states over 2 ranks, results come back in order on every rank
    states = list(range(5))
    outs = D.sharded_states(states, lambda s: torch.full((2, 2), float(s * s)))
    assert [float(o[0, 0]) for o in outs] == [0.0, 1.0, 4.0, 9.0, 16.0]
    # uneven slabs without counts
    g = D.all_gather_slabs(torch.full((rank + 1, 2), float(rank)))
    assert g.shape == (3, 2) and g[:, 0].tolist() == [0.0, 1.0, 1.0]
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
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)]
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
