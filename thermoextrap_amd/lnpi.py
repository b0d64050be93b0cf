r"""Beta expansion of the macrostate distribution ln Pi(N), the thermoextrap.lnpi
API (reference lnpi.py:42-170, 180-289, 373-438).

    (ln Pi)_energy = ln Pi - beta mu.N        d/dbeta lnPi = mu.N - <u>
so the series is  lnPi0,  mu.N - <u>,  -<u>',  -<u>'', ...  with <u> expanded by
the beta rules (central: d<u> = -du(2); raw: d u(1) = -u(2) + u(1)^2).
"""

from __future__ import annotations

from functools import lru_cache

import numpy as np
import torch

from . import beta as beta_xpan
from . import engine
from . import symbolic as S
from .data import DataCallbackABC
from .models import Derivatives, ExtrapModel
from .xrlite import DataArray, as_labelled, is_labelled


def _lnpi_rule(central: bool):
    def rule(atom):
        if atom[0] == "lnPi0":
            return S.Poly.atom(("mudotN",)) - (S.umean() if central else S.u_raw(1))
        if atom[0] == "mudotN":
            return S.Poly.const(0)
        return S.beta_rule(atom)

    return rule


@lru_cache(5)
def factory_derivatives(name="lnPi", n=None, d=None, xalpha=False, central=False, expand=True, post_func=None):
    """Expansion of ln(Pi/Pi_0); any other ``name`` defers to beta.factory_derivatives."""
    if name == "lnPi":
        series = beta_xpan.SymDerivBase(S.Poly.atom(("lnPi0",)), post_func=post_func, rule=_lnpi_rule(bool(central)),
                                        args=(("u", "du") if central else ("u",)) + ("lnPi0", "mudotN"))
        return Derivatives.from_series(series, args=series.args)
    return beta_xpan.factory_derivatives(name=name, n=n, d=d, xalpha=xalpha, central=central, post_func=post_func,
                                         expand=expand)


class lnPiDataCallback(DataCallbackABC):  # noqa: N801
    """Supplies ``lnPi0`` and ``mu . N`` to the derivative functions (reference lnpi.py:180-289)."""

    _fields = ("lnPi0", "mu", "dims_n", "dims_comp", "ncoords", "allow_resample")

    def __init__(self, lnPi0, mu, dims_n, dims_comp, ncoords=None, allow_resample=False):  # noqa: N803
        if not is_labelled(lnPi0):
            raise TypeError("lnPi0 must be a DataArray")
        if not is_labelled(mu):
            raise TypeError("mu must be a DataArray")
        self.lnPi0 = as_labelled(lnPi0)
        self.mu = as_labelled(mu)
        self.dims_n = (dims_n,) if isinstance(dims_n, str) else tuple(dims_n)
        self.dims_comp = dims_comp
        if ncoords is None:
            grid = np.meshgrid(*tuple(self.lnPi0[x].values for x in self.dims_n), indexing="ij")
            ncoords = DataArray(np.array(grid), (self.dims_comp, *self.dims_n))
        self.ncoords = as_labelled(ncoords)
        self.allow_resample = allow_resample
        self._cache = {}

    def check(self, data) -> None:
        pass

    @property
    def lnPi0_ave(self):  # noqa: N802
        return self.lnPi0

    @property
    def mudotN(self):  # noqa: N802
        """dot(mu, ncoords) over the component dim."""
        if "mudotN" not in self._cache:
            self._cache["mudotN"] = (self.mu * self.ncoords).sum(self.dims_comp)
        return self._cache["mudotN"]

    def resample(self, data, meta_kws=None, **kws):
        if not self.allow_resample:
            raise ValueError(
                "Must set `self.allow_resample` to `True` to use resampling. "
                "Resampling here is handled in an ad-hoc way, and should be used with care."
            )
        raise NotImplementedError("ad-hoc resampling of lnPi0 (reference lnpi.py:258-286) is out of scope")

    def derivs_args(self, data, derivs_args):
        return (*tuple(derivs_args), self.lnPi0_ave, self.mudotN)

    def device_sources(self, data, src, srcs):
        vdims = [d for d in src.out_dims if d in self.lnPi0.dims]
        ln = engine.to_device(np.ascontiguousarray(self.lnPi0.transpose(*vdims).values).ravel())
        mn = engine.to_device(np.ascontiguousarray(self.mudotN.transpose(*vdims).values).ravel())
        if ln.numel() != src.nval:
            raise ValueError("lnPi0 does not match the value dims of the data")
        i_ln, i_mn = len(srcs), len(srcs) + 1
        srcs.extend([ln, mn])
        return {"lnPi0": lambda a: (i_ln, 0, 0, 1), "mudotN": lambda a: (i_mn, 0, 0, 1)}


def factory_extrapmodel_lnPi(beta, data, *, central=None, order=None, alpha_name="beta", derivatives=None,  # noqa: N802
                             post_func=None, derivatives_kws=None):
    """ExtrapModel for lnPi; ``data`` must be ``x_is_u`` and carry a lnPiDataCallback."""
    if central is None:
        central = data.central
    if order is None:
        order = data.order + 1
    if central != data.central:
        raise ValueError
    if order > data.order + 1:
        raise ValueError
    if not data.x_is_u:
        raise ValueError
    if derivatives is None:
        derivatives = factory_derivatives(name="lnPi", central=central, post_func=post_func,
                                          **(derivatives_kws or {}))
    return ExtrapModel(alpha0=beta, data=data, derivatives=derivatives, order=order, alpha_name=alpha_name)
