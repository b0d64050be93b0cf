"""GP-input builder: the `(x_data, y_data, cov_data)` contract of
``thermoextrap.gpr_active.active_utils.input_GP_from_state`` (reference
gpr_active/active_utils.py:58-142), SURVEY 8(f) rank 2.  GPflow stays external;
this produces exactly what it consumes.

y_data = derivatives of the state (order+1, n_out); cov_data[k] = covariance of the
orders over bootstrap replicates for output k (``np.cov`` in the reference; here
txm_cov_over_rep on the device, on the replicate derivatives where txm_eval_poly left them).
"""

from __future__ import annotations

import math

import numpy as np

from . import engine
from .data import DataCentralMomentsVals


def _partial_bell(n: int, k: int, x) -> float:
    """Partial (exponential) Bell polynomial B_{n,k}(x_1, ..., x_{n-k+1})."""
    B = [[0.0] * (k + 1) for _ in range(n + 1)]
    B[0][0] = 1.0
    for nn in range(1, n + 1):
        for kk in range(1, min(nn, k) + 1):
            acc = 0.0
            for i in range(1, nn - kk + 2):
                acc += math.comb(nn - 1, i - 1) * x[i - 1] * B[nn - i][kk - 1]
            B[nn][kk] = acc
    return B[n][k]


def log_scale_matrix(alpha0: float, order: int) -> np.ndarray:
    """T with d^n y / d(log10 a)^n = sum_k T[n, k] d^k y / d a^k (Faa di Bruno;
    reference active_utils.py:108-126)."""
    T = np.zeros((order + 1, order + 1))
    T[0, 0] = 1.0
    for n in range(1, order + 1):
        for k in range(1, n + 1):
            xs = alpha0 * (np.log(10.0) ** np.arange(1, n + 1))  # x_i beyond n-k+1 do not enter B_{n,k}
            T[n, k] = _partial_bell(n, k, xs)
    return T


def input_GP_from_state(state, n_rep=100, log_scale=False, sampler=None):  # noqa: N802
    """(x_data, y_data, cov_data) for one ExtrapModel state.

    ``sampler`` overrides the default ``{"nrep": n_rep}`` (e.g. to fix a seed or
    force the device sampler)."""
    order = state.order
    alphas = state.alpha0 * np.ones((order + 1, 1))
    if log_scale:
        alphas = np.log10(alphas)
    x_data = np.concatenate([alphas, np.arange(order + 1)[:, None]], axis=1)

    if isinstance(state.data, DataCentralMomentsVals):
        derivs = state.derivs(norm=False).values
        boot = state.resample(sampler=sampler if sampler is not None else {"nrep": n_rep})
        vals, src = boot.derivatives.derivs(data=boot.data, order=order, norm=False, minus_log=boot.minus_log,
                                            _device=True)
    else:
        # pre-computed records: variance along 'rec' without resampling (reference lines 99-107)
        d = state.derivs(norm=False)
        derivs = d.mean("rec").values
        vals, src = state.derivatives.derivs(data=state.data, order=order, norm=False, minus_log=state.minus_log,
                                             _device=True)
    derivs = derivs.reshape(order + 1, -1)
    cov = engine.cov_over_rep(vals).cpu().numpy()  # (n_out, order+1, order+1)
    if log_scale:
        T = log_scale_matrix(state.alpha0, order)
        derivs = T @ derivs
        cov = np.einsum("ab,kbc,dc->kad", T, cov, T)
    return x_data, derivs, cov


def _batched_blocks(coll, spec, order, state0=0):
    """Bootstrap, derivatives and covariance over replicates of the states of ``coll`` (one shape), each ONE launch:
    -> (cov [S, n_out, n_ord, n_ord], derivs of the un-resampled states [S, n_ord, n_out]) on the device.
    ``state0``: index of coll[0] in the whole collection (its states draw stream replicates (state0 + s) * nrep ...)."""
    import torch

    from . import moments as cm

    S = len(coll)
    boot = coll.resample(spec, batched=True, state0=state0) if (state0 or S == 1) else coll.resample(spec)
    if getattr(boot, "_batch", None) is None:
        raise ValueError("the one-launch path needs states of one shape and a {'nrep': n, ...} sampler mapping")
    vals, _ = boot._derivs_batched(order=order, norm=False, _device=True)      # (order+1, S, nrep, n_out)
    n_ord, _, nrep, n_out = vals.shape
    cov_d = engine.cov_over_rep(vals.permute(0, 2, 1, 3).reshape(n_ord, nrep, S * n_out))  # (S*n_out, n_ord, n_ord)
    # the states' own derivatives: one evaluation over the stacked un-resampled states -- launched before anything is
    # copied back, so that the small copies of the caller are the step's only synchronisation points
    st0 = coll[0]
    stack = torch.stack([st.data.dxduave.device_values for st in coll])
    d0 = st0.data
    one = d0.new_like(dxduave=cm.CentralMomentsData(stack, mom_ndim=2, dims=("rep", *d0.dxduave.dims)), rec_dim="rep")
    dv_d, _ = st0.derivatives.derivs(data=one, order=order, norm=False, minus_log=st0.minus_log, _device=True)
    return cov_d.reshape(S, n_out, n_ord, n_ord), dv_d.permute(1, 0, 2).contiguous()


def _assemble(alpha0, y_s, cov, log_scale, order):
    """(alpha0 [S], y_s [S, n_ord, n_out], cov [S, n_out, n_ord, n_ord]) -> the `data_input` tuple create_GPR stacks
    (gpr_active/active_utils.py:896-925), without a Python loop over the states."""
    S, n_ord, n_out = y_s.shape
    if log_scale:
        T = np.stack([log_scale_matrix(a, order) for a in alpha0])                # (S, n_ord, n_ord)
        y_s = np.einsum("sab,sbk->sak", T, y_s)
        cov = np.einsum("sab,skbc,sdc->skad", T, cov, T)
        alpha0 = np.log10(alpha0)
    x_all = np.stack([np.repeat(alpha0, n_ord), np.tile(np.arange(n_ord, dtype=float), S)], axis=1)
    y_all = np.ascontiguousarray(y_s.reshape(S * n_ord, n_out))
    cov_all = np.zeros((n_out, S, n_ord, S, n_ord))
    idx = np.arange(S)
    cov_all[:, idx, :, idx, :] = cov                                              # block diagonal: (S, n_out, n_ord, n_ord)
    return x_all, y_all, cov_all.reshape(n_out, S * n_ord, S * n_ord)


def _input_gp_sharded(coll, n_rep, log_scale, spec, local):
    """State points over the ranks of an initialised torch.distributed group (BASELINE config 5 is quoted "8xMI355X";
    the reference loops over the states serially: gpr_active/active_utils.py:896-925, models.py:635-641).  Rank r runs
    the one-launch path on ITS states -- state i of the whole collection draws stream replicates i * nrep ... of one
    seed on whichever rank owns it (txm_sampler_spec.rep0) -- and ONE all-gather of the per-state blocks (alpha0,
    derivatives, covariance: (1 + n_ord n_out + n_out n_ord^2) doubles per state) ends the step: every rank returns the
    full tuple, bit for bit what one rank computes for the whole collection.  ``local``: ``coll`` holds only this
    rank's states (ranks in order); otherwise every rank passes the whole collection and takes its contiguous share."""
    import torch

    from . import distributed as D, moments as cm
    from .models import StateCollection

    rank, w = D.world()
    is_spec = isinstance(spec, dict) and "nrep" in spec and "indices" not in spec and "freq" not in spec
    if not is_spec or spec.get("device") is False:
        raise ValueError('sharded=... needs a {"nrep": n, ...} sampler mapping on the device stream (one stream for the '
                         "whole collection; numpy draws cannot be split over ranks consistently)")
    spec = {**spec, "device": True}
    if spec.get("seed") is None:
        spec["seed"] = D.broadcast_int(int(cm.validate_rng(spec.get("rng")).integers(0, 2**63 - 1)) if rank == 0 else 0)
    if local:
        counts = D.all_gather_ints(len(coll))
        mine = coll
    else:
        counts = D.shard_counts(len(coll), w)
        share = D.shard_range(len(coll), rank, w)
        mine = StateCollection(list(coll.states[share.start:share.stop]), kws=coll.kws)
    # Every decision that ends in a raise is taken on EVERY rank from gathered words: a rank that raised on a local condition
    # (no states, states of another shape) while the others went on into the all-gather below would leave them blocked in the
    # collective until its timeout.
    key = mine._batch_eligible(1) if len(mine) else None
    # a rank-comparable signature of the states' shape (the eligibility key holds a per-process object id)
    import zlib
    sig = -1 if key is None else zlib.crc32(repr((key[0][1:], key[1:10])).encode())
    sigs = D.all_gather_ints(sig)
    if min(counts) == 0:
        raise ValueError(f"more ranks than states: give every rank at least one state (states per rank: {list(counts)})")
    if min(sigs) < 0:
        raise ValueError("sharded=... needs ExtrapModel states over DataCentralMomentsVals of one shape "
                         f"(ranks without such states: {[r for r, v in enumerate(sigs) if v < 0]})")
    if len(set(sigs)) != 1:
        raise ValueError("sharded=...: the ranks hold states of different shapes / orders; the blocks cannot be gathered")
    state0 = sum(counts[:rank])
    order = mine.order
    cov_d, dv_d = _batched_blocks(mine, spec, order, state0=state0)
    S_loc, n_out, n_ord, _ = cov_d.shape
    a0 = torch.tensor([float(st.alpha0) for st in mine], dtype=torch.float64, device=cov_d.device)
    block = torch.cat([a0[:, None], dv_d.reshape(S_loc, -1), cov_d.reshape(S_loc, -1)], dim=1)
    full = D.all_gather_slabs(block, counts).cpu().numpy()                       # (S, 1 + n_ord n_out + n_out n_ord^2)
    S = full.shape[0]
    y_s = full[:, 1:1 + n_ord * n_out].reshape(S, n_ord, n_out)
    cov = full[:, 1 + n_ord * n_out:].reshape(S, n_out, n_ord, n_ord)
    return _assemble(full[:, 0].copy(), y_s, cov, log_scale, order)


def input_GP_from_states(states, n_rep=100, log_scale=False, sampler=None, sharded=False):  # noqa: N802
    """`input_GP_from_state` for a whole StateCollection (BASELINE config 5: 64 state points) with the
    bootstrap, the derivative evaluation and the covariance over replicates each done in ONE launch for all
    states.  The reference loops over the states calling input_GP_from_state and stacks the pieces
    (create_GPR, gpr_active/active_utils.py:896-925: np.vstack of x and y, scipy block_diag of the per-state
    covariances for every output); this returns exactly that `data_input` tuple:
    x_data (S * (order+1), 2), y_data (S * (order+1), n_out), noise_cov_mat (n_out, S*(order+1), S*(order+1)).
    Falls back to the per-state function when the states do not share a shape.

    ``sharded`` (extension): True -- every rank of the torch.distributed group passes the WHOLE collection and works on
    its contiguous share of the states; "local" -- every rank passes only ITS states (rank order = state order).  One
    all-gather of the per-state blocks; every rank returns the full tuple (see _input_gp_sharded)."""
    from .models import StateCollection

    coll = states if isinstance(states, StateCollection) else StateCollection(list(states))
    spec = sampler if sampler is not None else {"nrep": n_rep}
    if sharded:
        if sharded not in (True, "local"):
            raise ValueError('sharded must be False, True or "local"')
        return _input_gp_sharded(coll, n_rep, log_scale, spec, local=(sharded == "local"))
    S = len(coll)
    # the one-launch path needs what StateCollection.resample batches: states of one shape and a {"nrep": n, ...}
    # mapping (an {"indices": ...} / {"freq": ...} mapping goes through the per-state loop there, and here)
    is_spec = isinstance(spec, dict) and "nrep" in spec and "indices" not in spec and "freq" not in spec
    if coll._batch_eligible() is None or not is_spec:
        parts = [input_GP_from_state(st, n_rep=n_rep, log_scale=log_scale, sampler=sampler) for st in coll]
        x_all = np.concatenate([p_[0] for p_ in parts], axis=0)
        y_all = np.concatenate([p_[1] for p_ in parts], axis=0)
        n_out = parts[0][2].shape[0]
        n_ord = parts[0][2].shape[1]
        cov_all = np.zeros((n_out, S * n_ord, S * n_ord))
        for s_, p_ in enumerate(parts):
            cov_all[:, s_ * n_ord:(s_ + 1) * n_ord, s_ * n_ord:(s_ + 1) * n_ord] = p_[2]
        return x_all, y_all, cov_all
    order = coll.order
    cov_d, dv_d = _batched_blocks(coll, spec, order)
    alpha0 = np.array([float(st.alpha0) for st in coll])
    return _assemble(alpha0, dv_d.cpu().numpy(), cov_d.cpu().numpy(), log_scale, order)
