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
