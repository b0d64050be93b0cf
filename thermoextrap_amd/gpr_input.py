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


def input_GP_from_states(states, n_rep=100, log_scale=False, sampler=None):  # noqa: N802
    """`input_GP_from_state` for a whole StateCollection (BASELINE config 5: 64 state points) with the
    bootstrap, the derivative evaluation and the covariance over replicates each done in ONE launch for all
    states.  The reference loops over the states calling input_GP_from_state and stacks the pieces
    (create_GPR, gpr_active/active_utils.py:896-925: np.vstack of x and y, scipy block_diag of the per-state
    covariances for every output); this returns exactly that `data_input` tuple:
    x_data (S * (order+1), 2), y_data (S * (order+1), n_out), noise_cov_mat (n_out, S*(order+1), S*(order+1)).
    Falls back to the per-state function when the states do not share a shape."""
    import torch

    from .models import StateCollection

    coll = states if isinstance(states, StateCollection) else StateCollection(list(states))
    spec = sampler if sampler is not None else {"nrep": n_rep}
    S = len(coll)
    # the one-launch path needs what StateCollection.resample batches: states of one shape and a {"nrep": n, ...}
    # mapping (an {"indices": ...} / {"freq": ...} mapping goes through the per-state loop there, and here)
    is_spec = isinstance(spec, dict) and "nrep" in spec and "indices" not in spec and "freq" not in spec
    boot = coll.resample(spec) if (coll._batch_eligible() is not None and is_spec) else None
    if boot is None or getattr(boot, "_batch", None) is None:
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
    vals, _ = boot._derivs_batched(order=order, norm=False, _device=True)      # (order+1, S, nrep, n_out)
    n_ord, _, nrep, n_out = vals.shape
    cov_d = engine.cov_over_rep(vals.permute(0, 2, 1, 3).reshape(n_ord, nrep, S * n_out))  # (S*n_out, n_ord, n_ord)
    # the states' own derivatives: one evaluation over the stacked un-resampled states -- launched before anything is
    # copied back, so that the two small copies below are the step's only synchronisation points
    st0 = coll[0]
    from . import moments as cm

    stack = torch.stack([st.data.dxduave.device_values for st in coll])
    d0 = st0.data
    one = d0.new_like(dxduave=cm.CentralMomentsData(stack, mom_ndim=2, dims=("rep", *d0.dxduave.dims)), rec_dim="rep")
    dv_d, _ = st0.derivatives.derivs(data=one, order=order, norm=False, minus_log=st0.minus_log, _device=True)
    cov = cov_d.reshape(S, n_out, n_ord, n_ord).cpu().numpy()
    dv = dv_d.cpu().numpy()                                                       # (order+1, S, n_out)
    # stacked outputs without a Python loop over the states (64 of them at config 5)
    alpha0 = np.array([float(st.alpha0) for st in coll])
    y_s = dv.transpose(1, 0, 2)                                                   # (S, order+1, n_out)
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
    cov_all = cov_all.reshape(n_out, S * n_ord, S * n_ord)
    return x_all, y_all, cov_all
