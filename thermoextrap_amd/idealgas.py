"""Analytic 1-D ideal gas in an external field: synthetic-data source and closed
forms used by the tests and benchmarks (reference idealgas.py:82-266, 395-421).

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
