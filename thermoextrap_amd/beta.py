"""Inverse-temperature (beta) extrapolation: the thermoextrap.beta API
(reference beta.py:269-666) on the polynomial engine of symbolic.py.

``factory_derivatives`` returns a :class:`~thermoextrap_amd.models.Derivatives`
whose orders are generated lazily by the beta recursion

    d/dB du(n)      = -du(n+1) + n du(n-1) du(2)
    d/dB dxdu(n[,d]) = -dxdu(n+1,d) + n dxdu(n-1,d) du(2) + dxdu(1,d) du(n) [+ dxdu(n,d+1)]
    d/dB x1[d]      = -dxdu(1,d) [+ x1[d+1]]           d/dB <u> = -du(2)
    d/dB u(n)       = -u(n+1) + u(n) u(1)
    d/dB xu(n[,d])  = -xu(n+1,d) + xu(n,d) u(1) [+ xu(n,d+1)]

(reference beta.py:52-54, 83-85, 110-116, 144-151, 174-176, 193-196, 216-218,
246-258) and evaluated on the device from the data object's moment states.
"""

from __future__ import annotations

from functools import lru_cache

from . import symbolic as S
from .models import Derivatives, ExtrapModel, PerturbModel, SymDerivBase

__all__ = ["SymDerivBeta", "factory_derivatives", "factory_extrapmodel", "factory_perturbmodel"]


class SymDerivBeta(SymDerivBase):
    r"""Symbolic expressions for :math:`d^n \langle \cdot \rangle / d\beta^n`."""

    beta = "beta"

    def __init__(self, func, args=None, expand=True, post_func=None):
        super().__init__(func, args=args, expand=expand, post_func=post_func, rule=S.beta_rule)

    # ---- constructors (same names/arguments as the reference) ---------------
    @classmethod
    def x_ave(cls, xalpha=False, central=None, expand=True, post_func=None):
        """<x>: central -> (x1, du, dxdu) symbols; raw -> (u, xu)."""
        if central:
            func = S.x1(0) if xalpha else S.x1()
            args = ("x1", "du", "dxdu")
        else:
            func = S.xu_raw(0, 0) if xalpha else S.xu_raw(0)
            args = ("u", "xu")
        return cls(func=func, args=args, expand=expand, post_func=post_func)

    @classmethod
    def u_ave(cls, central=None, expand=True, post_func=None):
        """<u>."""
        if central:
            return cls(func=S.umean(), args=("u", "du"), expand=expand, post_func=post_func)
        return cls(func=S.u_raw(1), args=("u",), expand=expand, post_func=post_func)

    @classmethod
    def dun_ave(cls, n, expand=True, post_func=None, central=None):
        """<(u - <u>)^n>, n > 1."""
        if central is not None and not central:
            raise ValueError(f"central={central} must be None or evaluate to True")
        if (n := int(n)) <= 1:
            raise ValueError(f"n={n} must be > 1.")
        return cls(func=S.du(n), args=("u", "du"), expand=expand, post_func=post_func)

    @classmethod
    def dxdun_ave(cls, n, xalpha=False, expand=True, post_func=None, d=None, central=None):
        """<dx^(d) du^n>, n > 0."""
        if central is not None and not central:
            raise ValueError(f"central={central} nust be `None` or evaluate to `True`")
        if (n := int(n)) <= 0:
            raise ValueError(f"n={n} must be positive integer.")
        if xalpha:
            if not isinstance(d, int):
                raise TypeError
            func = S.dxdu(n, d)
        else:
            func = S.dxdu(n)
        return cls(func=func, args=("x1", "du", "dxdu"), expand=expand, post_func=post_func)

    @classmethod
    def un_ave(cls, n, expand=True, post_func=None, central=None):
        """<u^n>, n >= 1."""
        if central is not None and central:
            raise ValueError(f"central={central} must be `None` or evaluate to False")
        if (n := int(n)) < 1:
            raise ValueError(f"n={n} must be >=1.")
        return cls(func=S.u_raw(n), args=("u",), expand=expand, post_func=post_func)

    @classmethod
    def xun_ave(cls, n, d=None, xalpha=False, expand=True, post_func=None, central=None):
        """<x^(d) u^n>, n >= 0."""
        if central is not None and central:
            raise ValueError(f"central={central} must be `None` or False")
        if (n := int(n)) < 0:
            raise ValueError(f"n={n} must be >= 0")
        if xalpha:
            if not isinstance(d, int):
                raise TypeError
            if d < 0:
                raise ValueError
            func = S.xu_raw(n, d)
        else:
            func = S.xu_raw(n)
        return cls(func=func, args=("u", "xu"), expand=expand, post_func=post_func)

    @classmethod
    def from_name(cls, name, xalpha=False, central=None, expand=True, post_func=None, n=None, d=None):
        func = getattr(cls, name, None)
        if func is None or name.startswith("_") or name in {"from_name", "expr"}:
            raise ValueError(f"{name} not found")
        kws = {"expand": expand, "post_func": post_func, "central": central}
        if name == "x_ave":
            kws.update(xalpha=xalpha)
        elif name in {"dun_ave", "un_ave"}:
            kws.update(n=n)
        elif name in {"dxdun_ave", "xun_ave"}:
            kws.update(n=n, xalpha=xalpha, d=d)
        return func(**kws)


@lru_cache(5)
def factory_derivatives(name="x_ave", n=None, d=None, xalpha=False, central=None, post_func=None, expand=True):
    """Derivatives object for the named average (reference beta.py:532-573).

    name : {x_ave, u_ave, dxdun_ave, dun_ave, un_ave, xun_ave}
    """
    derivs = SymDerivBeta.from_name(name=name, n=n, d=d, xalpha=xalpha, central=central, post_func=post_func,
                                    expand=expand)
    return Derivatives.from_series(derivs, args=derivs.args)


def factory_extrapmodel(beta, data, *, name="x_ave", n=None, d=None, xalpha=None, central=None, order=None,
                        alpha_name="beta", derivatives=None, post_func=None, derivatives_kws=None):
    """ExtrapModel for a beta expansion; ``order``, ``xalpha`` and ``central``
    default to the data object's (reference beta.py:577-666)."""
    if xalpha is None:
        xalpha = data.xalpha
    if central is None:
        central = data.central
    if order is None:
        order = data.order
    if xalpha != data.xalpha:
        raise ValueError(f"xalpha={xalpha} must equal data.xalpha={data.xalpha}")
    if central != data.central:
        raise ValueError(f"central={central} must equal data.central={data.central}")
    if order > data.order:
        raise ValueError(f"order={order} must be <= data.order={data.order}")
    if derivatives is None:
        if name in {"u_ave", "un_ave", "dun_ave"} and not data.x_is_u:
            raise ValueError("if name in [u_ave, un_ave, dun_ave] must have data.x_is_u")
        derivatives = factory_derivatives(name=name, n=n, d=d, xalpha=xalpha, central=central, post_func=post_func,
                                          **(derivatives_kws or {}))
    return ExtrapModel(alpha0=beta, data=data, derivatives=derivatives, order=order, alpha_name=alpha_name)


def factory_perturbmodel(beta, uv, xv, alpha_name="beta", **kws):
    """PerturbModel for a beta expansion (reference beta.py:669-696)."""
    from .data import DataValues

    data = DataValues.from_vals(xv=xv, uv=uv, order=0, **kws)
    return PerturbModel(alpha0=beta, data=data, alpha_name=alpha_name)
