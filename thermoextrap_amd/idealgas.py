"""Analytic 1-D ideal gas in an external field: synthetic-data source and closed
forms used by the tests and benchmarks (reference idealgas.py:82-421: moments, densities, samplers, the analytic
beta- and volume-derivatives and the Taylor extrapolations built from them).

Single particle on [0, L] with potential u = x:  p(x) ~ exp(-beta x).
"""

from __future__ import annotations

import math
from functools import lru_cache

import numpy as np

from .moments import validate_rng


def x_ave(beta, vol=1.0):
    """<x> = 1/beta - L / (exp(beta L) - 1)."""
    return 1.0 / beta - vol / (np.exp(beta * vol) - 1.0)


def x_var(beta, vol=1.0):
    """Var(x) = 1/beta^2 - L^2 e^{beta L} / (e^{beta L} - 1)^2  (= -d<x>/dbeta; reference idealgas.py:101-113)."""
    e = np.exp(beta * vol)
    return 1.0 / beta**2 - vol**2 * e / (e - 1.0) ** 2


def x_prob(x, beta, vol=1.0):
    """Canonical density of one particle's position on [0, L]: beta e^{-beta x} / (1 - e^{-beta L})
    (reference idealgas.py:116-127)."""
    return beta * np.exp(-beta * x) / (1.0 - np.exp(-beta * vol))


def x_cdf(x, beta, vol=1.0):
    """Cumulative distribution of the position (what x_sample inverts; reference idealgas.py:152-163)."""
    return (1.0 - np.exp(-beta * x)) / (1.0 - np.exp(-beta * vol))


def u_prob(u, npart, beta, vol=1.0):
    """Large-N density of the total potential energy of ``npart`` particles: Normal(npart <x>, npart Var(x))
    (reference idealgas.py:131-147)."""
    mean = npart * x_ave(beta, vol)
    std = np.sqrt(npart * x_var(beta, vol))
    z = (u - mean) / std
    return np.exp(-0.5 * z * z) / (std * np.sqrt(2.0 * np.pi))


def x_sample(shape, beta, vol=1.0, rng=None):
    """Positions by inversion of the CDF (1 - e^{-beta x}) / (1 - e^{-beta L})."""
    r = validate_rng(rng).random(shape)
    return (-1.0 / beta) * np.log(1.0 - r * (1.0 - np.exp(-beta * vol)))


def u_sample(shape, beta, vol=1.0, rng=None):
    return x_sample(shape=shape, beta=beta, vol=vol, rng=rng).sum(axis=-1)


def generate_data(shape, beta, vol=1.0, rng=None):
    """(x, u): mean position and total potential energy of ``shape[1]``
    independent particles for each of ``shape[0]`` configurations."""
    positions = x_sample(shape=shape, beta=beta, vol=vol, rng=rng)
    return positions.mean(axis=-1), positions.sum(axis=-1)


@lru_cache(maxsize=100)
def _dbeta(k, kind):
    import sympy as sp

    b, L = sp.symbols("b L", positive=True)
    xave = 1 / b - L / (sp.exp(b * L) - 1)
    if kind == "vol":  # d^k <x> / dL^k
        return sp.lambdify([b, L], sp.diff(xave, L, k), "numpy")
    f = {"xave": xave, "minuslog": -sp.log(xave), "depend": b * xave, "depend_minuslog": -sp.log(b * xave)}[kind]
    return sp.lambdify([b, L], sp.diff(f, b, k), "numpy")


def dbeta_xave(k):
    """k-th beta-derivative of <x>."""
    return _dbeta(k, "xave")


def dbeta_xave_minuslog(k):
    return _dbeta(k, "minuslog")


def dbeta_xave_depend(k):
    """k-th beta-derivative of <beta x>."""
    return _dbeta(k, "depend")


def dbeta_xave_depend_minuslog(k):
    return _dbeta(k, "depend_minuslog")


def _extrap(fn, order, beta0, beta, vol):
    dbeta = np.asarray(beta) - beta0
    out, tot = [], np.zeros_like(dbeta, dtype=float)
    for k in range(order + 1):
        val = fn(k)(beta0, vol)
        out.append(val)
        tot = tot + val * dbeta**k / math.factorial(k)
    return tot, np.array(out)


def x_beta_extrap(order, beta0, beta, vol=1.0):
    """(Taylor prediction at beta, exact derivatives at beta0) for <x>."""
    return _extrap(dbeta_xave, order, beta0, beta, vol)


def x_beta_extrap_minuslog(order, beta0, beta, vol=1.0):
    return _extrap(dbeta_xave_minuslog, order, beta0, beta, vol)


def x_beta_extrap_depend(order, beta0, beta, vol=1.0):
    return _extrap(dbeta_xave_depend, order, beta0, beta, vol)


def x_beta_extrap_depend_minuslog(order, beta0, beta, vol=1.0):
    return _extrap(dbeta_xave_depend_minuslog, order, beta0, beta, vol)


def dvol_xave(k):
    """k-th derivative of <x> with respect to the box length L, as a function of (beta, L) (reference idealgas.py:259-266)."""
    return _dbeta(k, "vol")


def x_vol_extrap(order, vol0, vol, beta=1.0):
    """(Taylor prediction at vol, exact L-derivatives at vol0) for <x> at fixed beta (reference idealgas.py:377-400)."""
    dvol = np.asarray(vol) - vol0
    out, tot = [], np.zeros_like(dvol, dtype=float)
    for k in range(order + 1):
        val = dvol_xave(k)(beta, vol0)
        out.append(val)
        tot = tot + val * dvol**k / math.factorial(k)
    return tot, np.array(out)
