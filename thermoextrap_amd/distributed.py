"""Multi-GPU sharding of the hot path (one process per GPU, torch.distributed;
backend "nccl" is RCCL over xGMI on MI355X, "gloo" in CPU tests).

The path shards with **no data-path collective** (SURVEY 8(e)): state points and
bootstrap replicates are independent.  Every rank works on its contiguous share
and one all-gather of the small result slabs ([nrep, C, 2, K] states or
[order+1, nrep, C] derivatives; <= a few MB) ends the step -- each GPU sends its
slab on all xGMI links at once; never a ring all-reduce of bulk data.
"""

from __future__ import annotations

from collections.abc import Callable, Sequence

import torch
import torch.distributed as dist


def world() -> tuple[int, int]:
    """(rank, world_size); (0, 1) when torch.distributed is not initialised."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(n: int, rank: int | None = None, world_size: int | None = None) -> range:
    """Contiguous share of ``range(n)``: the first ``n % world`` ranks get one extra item."""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    if not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} outside [0, {world_size})")
    base, extra = divmod(n, world_size)
    lo = rank * base + min(rank, extra)
    return range(lo, lo + base + (1 if rank < extra else 0))


def shard_counts(n: int, world_size: int) -> list[int]:
    return [len(shard_range(n, r, world_size)) for r in range(world_size)]


def all_gather_slabs(local: torch.Tensor, counts: Sequence[int] | None = None, group=None) -> torch.Tensor:
    """Concatenate every rank's slab along dim 0 (slabs may differ in dim-0 length)."""
    rank, w = world()
    if w == 1:
        return local
    if local.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal on a box without RCCL peers: gather through host memory
        return all_gather_slabs(local.cpu(), counts, group).to(local.device)
    if counts is None:
        n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
        ns = [torch.zeros_like(n) for _ in range(w)]
        dist.all_gather(ns, n, group=group)
        counts = [int(t.item()) for t in ns]
    nmax = max(counts)
    pad = local
    if local.shape[0] < nmax:  # equal-size buffers keep the collective a plain all-gather
        pad = torch.zeros((nmax, *local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(w)]
    dist.all_gather(bufs, pad.contiguous(), group=group)
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)


def all_gather_ints(value: int, group=None) -> list[int]:
    """Every rank's Python int, in rank order."""
    rank, w = world()
    if w == 1:
        return [int(value)]
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.tensor([int(value)], dtype=torch.int64, device=dev)
    ts = [torch.zeros_like(t) for _ in range(w)]
    dist.all_gather(ts, t, group=group)
    return [int(v.item()) for v in ts]


def broadcast_int(value: int, src: int = 0, group=None) -> int:
    """The same Python int on every rank (rank `src`'s)."""
    rank, w = world()
    if w == 1:
        return int(value)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.tensor([int(value)], dtype=torch.int64, device=dev)
    dist.broadcast(t, src=src, group=group)
    return int(t.item())


def broadcast_tensor(t: torch.Tensor, src: int = 0, group=None) -> torch.Tensor:
    """Rank ``src``'s values of ``t`` on every rank (in place for the others; returns ``t``)."""
    rank, w = world()
    if w == 1:
        return t
    if t.is_cuda and dist.get_backend(group) == "gloo":  # rehearsal on a box without RCCL peers: through host memory
        h = t.cpu()
        dist.broadcast(h, src=src, group=group)
        t.copy_(h)
        return t
    dist.broadcast(t, src=src, group=group)
    return t


def sharded_reduce(x: torch.Tensor, u: torch.Tensor, order: int, w: torch.Tensor | None = None, group=None, ops=None) -> torch.Tensor:
    """SAMPLE-sharded reduce (SURVEY 8(e) partition (4)): every rank holds a shard ``x [N_r, C]``, ``u [N_r]`` (``w``) of one
    long sample array -- the plain reduce is the HBM-bound leg of the path, and N / world samples per rank put every
    GPU's HBM stacks to work.  Rank 0's pivot estimate is broadcast, every rank takes the power sums of ITS samples about
    that pivot (they add exactly like the samples), ONE all-gather of the ``[C, 2, K]`` sums (a few KB) and a fixed-order
    addition + shift give every rank the same state, bit for bit -- the state of the concatenated samples
    (``cmomy.wrap_reduce_vals`` over the whole array: reference data.py:1632-1640) to rounding.  No bulk data moves.

    ``ops``: (pivot, sums, finish) callables standing in for the engine's (the gloo tests run the collective logic on the CPU)."""
    from . import engine

    pivot_fn, sums_fn, finish_fn = ops if ops is not None else (
        engine.reduce_pivot, lambda x_, u_, o_, p_, w_: engine.reduce_sums(x_, u_, o_, p_, w=w_), engine.sums_to_state)
    # shard sizes first: a rank with an EMPTY shard (more ranks than samples, an uneven loader) must neither fail locally
    # on the library's N >= 1 -- the other ranks would then block in the broadcast below until the timeout -- nor supply
    # the pivot.  Every decision below is made from the gathered sizes, i.e. identically on every rank.
    sizes = all_gather_ints(int(x.shape[0]), group)
    if not any(sizes):
        raise ValueError("sharded_reduce: every rank's shard is empty")
    src = next(r for r, n in enumerate(sizes) if n > 0)     # the first rank that holds samples estimates the pivot
    rank, nw = world()
    C = 1 if x.dim() == 1 else int(x.shape[1])
    if sizes[rank] > 0:
        piv = pivot_fn(x, u) if rank == src else torch.empty(1 + C, dtype=torch.float64, device=x.device)
    else:
        piv = torch.empty(1 + C, dtype=torch.float64, device=x.device)
    piv = broadcast_tensor(piv, src, group)
    if sizes[rank] > 0:
        sums = sums_fn(x, u, order, piv, w)
    else:                                                   # an empty shard adds nothing: zero sums of the common shape
        sums = torch.zeros((C, 2, order + 1), dtype=torch.float64, device=x.device)
    stack = all_gather_slabs(sums.unsqueeze(0), None if nw == 1 else [1] * nw, group)
    return finish_fn(stack, piv)


def replicate_offsets(nrep: int, world_size: int) -> list[int]:
    """First stream replicate of every rank's slab: rank r computes replicates
    ``[offsets[r], offsets[r] + counts[r])`` of ONE stream (one seed), so the gathered result is the one-rank
    result bit for bit (txm_sampler_spec.rep0)."""
    counts = shard_counts(nrep, world_size)
    offs, a = [], 0
    for c in counts:
        offs.append(a)
        a += c
    return offs


def sharded_bootstrap(compute: Callable[[int, int, int], torch.Tensor], nrep: int, seed: int, group=None,
                      rep0: int = 0) -> torch.Tensor:
    """Replicate-sharded bootstrap of ONE state point whose samples every rank holds:
    rank r computes ``compute(nrep_r, seed, rep0_r)`` -> slab ``[nrep_r, ...]`` (replicates ``rep0_r ...`` of the
    stream of ``seed``) and all ranks receive the ``[nrep, ...]`` concatenation -- identical, bit for bit, to
    ``compute(nrep, seed, rep0)`` on one rank."""
    rank, w = world()
    counts = shard_counts(nrep, w)
    slab = compute(counts[rank], seed, rep0 + replicate_offsets(nrep, w)[rank])
    if slab.shape[0] != counts[rank]:
        raise ValueError("compute returned the wrong number of replicates")
    return all_gather_slabs(slab, counts, group)


def sharded_states(states: Sequence, func: Callable, group=None) -> list:
    """State-point sharding (reference models.py:635-641 runs this loop serially):
    rank r evaluates ``func(state)`` -> tensor for its share of ``states``; every
    rank gets the full list back, in order.  All results must share one shape."""
    rank, w = world()
    mine = shard_range(len(states), rank, w)
    outs = [func(states[i]) for i in mine]
    if w == 1:
        return outs
    counts = shard_counts(len(states), w)
    if not outs:
        raise ValueError("more ranks than states: give every rank at least one state")
    full = all_gather_slabs(torch.stack(outs), counts, group)
    return list(full.unbind(0))


def run_step(mode: str, compute: Callable[[int, int, int], torch.Tensor], nrep: int, seed: int, group=None) -> torch.Tensor:
    """One multi-GPU bootstrap step, the way bench.py (and a user script) shards it -- the single place both the
    benchmark and the gloo tests go through.  ``compute(n, seed, rep0)`` returns the ``[n, ...]`` slab of stream
    replicates ``rep0 .. rep0 + n`` of ``seed``.

    "states"   every rank bootstraps ITS OWN state point: state ``rank`` of a collection that shares one seed draws
               replicates ``rank * nrep ...`` (independent across states, as StateCollection.resample does);
               returns the ``[world * nrep, ...]`` concatenation of all ranks' slabs (weak scaling).
    "replicas" every rank holds the SAME state point and computes its contiguous ``nrep / world`` replicates of the
               one stream (`sharded_bootstrap`); returns the ``[nrep, ...]`` concatenation (strong scaling) --
               bit for bit what one rank computes for all ``nrep``.
    One all-gather ends the step; there is no data-path collective."""
    rank, w = world()
    if mode == "replicas":
        return sharded_bootstrap(compute, nrep, seed, group)
    if mode != "states":
        raise ValueError(f"unknown mode {mode!r}")
    slab = compute(nrep, seed, rank * nrep)
    if slab.shape[0] != nrep:
        raise ValueError("compute returned the wrong number of replicates")
    return all_gather_slabs(slab, [nrep] * w, group)
