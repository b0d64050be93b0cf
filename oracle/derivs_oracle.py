"""
oracle/derivs_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT.

Independent numerical restatement of the derivative formulas the reference
generates symbolically (src/thermoextrap/beta.py:32-266 -> models.py:317-383).
No symbolic algebra here: the same quantities are obtained from the identity
the reference's *legacy* oracle is built on (legacy/utilities.py:61-111,
180-245):

    <a>(beta) = f / z,   f = sum_i a_i e^{-beta u_i},   z = sum_i e^{-beta u_i}
    f^(k) / z = (-1)^k <a u^k>             z^(k) / z = (-1)^k <u^k>
    with a = a(beta):   f^(n) / z = sum_k C(n,k) (-1)^k <a^(n-k) u^k>

and the general Leibniz rule for a quotient,
    g = f / z   =>   g^(n) = f^(n)/z - sum_{k<n} C(n,k) g^(k) z^(n-k)/z ,
evaluated with truncated Taylor "jets" so that products, powers and -log of
averages (central moments, post_func, minus_log) follow by jet arithmetic.

Pinned against tests/golden/fixture_legacy.npz (outputs of the reference's own
legacy code) in tests/test_derivs_oracle.py.
"""

from __future__ import annotations

import math

import numpy as np


# ---------------------------------------------------------------------------
# jets: arrays whose leading axis holds derivatives d^k/dbeta^k, k = 0..n
# ---------------------------------------------------------------------------
def jet_mul(a, b):
    n = a.shape[0]
    out = np.zeros(np.broadcast_shapes(a.shape, b.shape), dtype=np.result_type(a, b))
    for k in range(n):
        for j in range(k + 1):
            out[k] = out[k] + math.comb(k, j) * a[j] * b[k - j]
    return out


def jet_pow(a, p: int):
    out = np.zeros_like(a)
    out[0] = 1.0
    for _ in range(p):
        out = jet_mul(out, a)
    return out


def jet_minus_log(a):
    """derivatives of -log(a) from those of a:  y' = -a'/a  =>  a y' = -a'."""
    n = a.shape[0]
    y = np.zeros_like(a)
    y[0] = -np.log(a[0])
    # (a * y')^(k-1) = -a^(k):  sum_j C(k-1,j) a^(j) y^(k-j) = -a^(k)
    for k in range(1, n):
        acc = -a[k]
        for j in range(1, k):
            acc = acc - math.comb(k - 1, j) * a[j] * y[k - j]
        y[k] = acc / a[0]
    return y


def average_jet(raw_au, raw_u, order):
    """Jet of <a>(beta) for a beta-independent sample function a.

    raw_au[k] = <a u^k> (k <= order; trailing dims free), raw_u[k] = <u^k> (scalars or
    broadcastable)."""
    g = [None] * (order + 1)
    for n in range(order + 1):
        acc = (-1) ** n * np.asarray(raw_au[n])
        for k in range(n):
            acc = acc - math.comb(n, k) * g[k] * (-1) ** (n - k) * raw_u[n - k]
        g[n] = acc
    return np.stack([np.broadcast_to(v, np.shape(g[-1])) for v in g])


def average_jet_xalpha(raw_xu_dk, raw_u, order):
    """Jet of <x(beta)>: raw_xu_dk[d][k] = <x^(d) u^k>."""
    g = [None] * (order + 1)
    for n in range(order + 1):
        fn = 0.0
        for k in range(n + 1):
            fn = fn + math.comb(n, k) * (-1) ** k * np.asarray(raw_xu_dk[n - k][k])
        acc = fn
        for k in range(n):
            acc = acc - math.comb(n, k) * g[k] * (-1) ** (n - k) * raw_u[n - k]
        g[n] = acc
    return np.stack(g)


# ---------------------------------------------------------------------------
# the named averages of beta.factory_derivatives
# ---------------------------------------------------------------------------
def raw_moments(x, u, kmax, w=None):
    """<u^k>, <x u^k> for k <= kmax by direct summation in extended precision."""
    x = np.asarray(x, dtype=np.longdouble)
    u = np.asarray(u, dtype=np.longdouble)
    w = np.ones_like(u) if w is None else np.asarray(w, dtype=np.longdouble)
    W = w.sum()
    xs = x.reshape(x.shape[0], -1)
    ru = np.empty(kmax + 1, dtype=np.longdouble)
    rxu = np.empty((kmax + 1, xs.shape[1]), dtype=np.longdouble)
    wuk = w.copy()  # w * u^k by repeated products (extended precision; the full-size tests run this on 1e8 samples)
    for k in range(kmax + 1):
        ru[k] = wuk.sum() / W
        for c in range(xs.shape[1]):
            rxu[k, c] = (wuk * xs[:, c]).sum() / W
        if k < kmax:
            wuk *= u
    # kept in extended precision: at order 6 the raw-moment recursion cancels ~13 digits
    # when u ~ 175 +- 5 (the ideal-gas notebook data); callers cast the final jets to float.
    return ru, rxu.reshape((kmax + 1,) + x.shape[1:])


def derivs_x_ave(x, u, order, w=None, post_func=None, minus_log=False):
    """d^n <x> / dbeta^n, n <= order; x: (N,) or (N, C).  Output (order+1, C)."""
    ru, rxu = raw_moments(x, u, order, w)
    j = average_jet(rxu, ru, order)
    if post_func == "minus_log" or minus_log:
        j = jet_minus_log(j)
    elif isinstance(post_func, str) and post_func.startswith("pow_"):
        j = jet_pow(j, int(post_func.split("_")[-1]))
    return np.asarray(j, dtype=float)


def derivs_x_ave_xalpha(xd, u, order, w=None, minus_log=False):
    """xd: (N, D, C) with xd[:, d] = d^d x / dbeta^d samples."""
    u = np.asarray(u, dtype=float)
    D = xd.shape[1]
    if D < order + 1:
        raise ValueError("need derivatives of x up to `order`")
    ru, _ = raw_moments(xd[:, 0], u, order, w)
    raw = [raw_moments(xd[:, d], u, order, w)[1] for d in range(order + 1)]
    j = average_jet_xalpha(raw, ru, order)
    return np.asarray(jet_minus_log(j) if minus_log else j, dtype=float)


def derivs_un_ave(u, n, order, w=None):
    """d^k <u^n> / dbeta^k."""
    ru, _ = raw_moments(u, u, order + n, w)
    return np.asarray(average_jet([ru[n + k] for k in range(order + 1)], ru, order), dtype=float)


def derivs_dun_ave(u, n, order, w=None):
    """d^k <(u-<u>)^n> / dbeta^k via jets of the raw moments."""
    ujets = [derivs_un_ave(u, m, order, w) if m > 0 else np.r_[1.0, np.zeros(order)] for m in range(n + 1)]
    mean = ujets[1]
    out = np.zeros(order + 1)
    for m in range(n + 1):
        out = out + math.comb(n, m) * jet_mul(ujets[m], jet_pow(-mean, n - m))
    return out


def predict(derivs, alpha0, alphas, order=None):
    """Taylor series sum_k derivs[k] (alpha-alpha0)^k / k!  -> (n_alpha, ...)."""
    derivs = np.asarray(derivs)
    order = derivs.shape[0] - 1 if order is None else order
    d = np.asarray(alphas, dtype=float) - alpha0
    out = 0.0
    for k in range(order + 1):
        out = out + np.multiply.outer(d**k, derivs[k]) / math.factorial(k)
    return out


def weighted_predict(derivs_pair, alpha0_pair, alphas, order=None, m=20):
    """Minkowski-weighted blend of two Taylor series (reference legacy/interp.py:71-126,
    models.py:726-728, 835-858): w_s = 1 - d_s^m / (d_0^m + d_1^m), out = sum w_s p_s / sum w_s."""
    alphas = np.atleast_1d(np.asarray(alphas, dtype=float))
    p = [predict(derivs_pair[s], alpha0_pair[s], alphas, order) for s in range(2)]
    d = [np.abs(alphas - alpha0_pair[s]) ** m for s in range(2)]
    w = [1.0 - d[s] / (d[0] + d[1]) for s in range(2)]
    shape = (-1,) + (1,) * (p[0].ndim - 1)
    return (p[0] * w[0].reshape(shape) + p[1] * w[1].reshape(shape)) / (w[0] + w[1]).reshape(shape)


def interp_coefs(derivs_states, alpha0_states):
    """Hermite interpolation polynomial through value + derivatives at each state
    (reference legacy/interp.py:157-235, models.py:861-925): coefficients c_p with
    sum_p c_p p!/(p-j)! a_s^(p-j) = derivs[s][j].  Solved EXACTLY in rational
    arithmetic on the given doubles (the reference inverts the matrix in float64;
    this is the checker, so it does not share that conditioning)."""
    from fractions import Fraction

    derivs_states = [np.asarray(d, dtype=float) for d in derivs_states]
    nord = derivs_states[0].shape[0]
    n = len(derivs_states) * nord
    tail = derivs_states[0].shape[1:]
    rhs = np.concatenate([d.reshape(nord, -1) for d in derivs_states], axis=0)
    ncol = rhs.shape[1]
    A = []
    for a0 in alpha0_states:
        fa = Fraction(float(a0))
        for j in range(nord):
            A.append([Fraction(math.perm(p, j)) * fa ** (p - j) if p >= j else Fraction(0) for p in range(n)])
    B = [[Fraction(float(v)) for v in row] for row in rhs]
    for c in range(n):                       # Gauss-Jordan, exact
        piv = next(r for r in range(c, n) if A[r][c] != 0)
        A[c], A[piv] = A[piv], A[c]
        B[c], B[piv] = B[piv], B[c]
        inv = 1 / A[c][c]
        A[c] = [v * inv for v in A[c]]
        B[c] = [v * inv for v in B[c]]
        for r in range(n):
            if r != c and A[r][c] != 0:
                f = A[r][c]
                A[r] = [x - f * y for x, y in zip(A[r], A[c])]
                B[r] = [x - f * y for x, y in zip(B[r], B[c])]
    out = np.array([[float(v) for v in row] for row in B]).reshape((n, *tail))
    exact = [[v for v in row] for row in B]
    return out, exact


def interp_predict(derivs_states, alpha0_states, alphas):
    """sum_p c_p alpha^p with the exact coefficients, rounded once at the end."""
    from fractions import Fraction

    _, exact = interp_coefs(derivs_states, alpha0_states)
    tail = np.asarray(derivs_states[0]).shape[1:]
    alphas = np.atleast_1d(np.asarray(alphas, dtype=float))
    out = np.empty((len(alphas), len(exact[0])))
    for i, a in enumerate(alphas):
        fa = Fraction(float(a))
        for c in range(len(exact[0])):
            out[i, c] = float(sum(exact[p][c] * fa**p for p in range(len(exact))))
    return out.reshape((len(alphas), *tail))
