"""Volume expansion of the 1-D ideal gas test system (reference
volume_idealgas.py:16-150):  d0 = <x>,  d1 = (<x W> - <x><W>) / L + <x> / L."""

from __future__ import annotations

from fractions import Fraction
from functools import lru_cache

from . import symbolic as S
from .models import Derivatives, ExtrapModel
from .volume import _FirstOrderSeries


@lru_cache(5)
def factory_derivatives(refV=1.0):  # noqa: N803
    inv = S.Poly.const(1 / Fraction(float(refV)))  # exact rational 1 / refV (refV itself is a dyadic rational)
    xw0, xw1, w1 = S.xu_raw(0), S.xu_raw(1), S.u_raw(1)
    return Derivatives(_FirstOrderSeries([xw0, (xw1 - xw0 * w1 + xw0) * inv]), args=("W", "xW"))


def factory_extrapmodel(volume, uv, xv, order=1, alpha_name="volume", **kws):
    if order != 1:
        raise ValueError("only first order supported")
    from .data import factory_data_values

    data = factory_data_values(uv=uv, xv=xv, order=order, central=False, xalpha=False, **kws)
    return ExtrapModel(alpha0=volume, data=data, derivatives=factory_derivatives(refV=volume), order=order,
                       minus_log=False, alpha_name=alpha_name)


def factory_extrapmodel_data(volume, data, order=1, alpha_name="volume"):
    if order is None:
        order = data.order
    if order != 1:
        raise ValueError("only first order supported")
    if order > data.order:
        raise ValueError
    if data.central:
        raise ValueError("Only works with raw moments.")
    if data.deriv_dim is not None:
        raise ValueError("Cannot include derivatives of observable.")
    return ExtrapModel(alpha0=volume, data=data, derivatives=factory_derivatives(refV=volume), order=order,
                       minus_log=False, alpha_name=alpha_name)
