"""Volume extrapolation (first order only), the thermoextrap.volume API
(reference volume.py:39-233).  Only DataValues-like objects are supported, as
in the reference.

    d0 = <x>            d1 = (-<x><W> + <x W> + <dxdq>) / (V ndim)

with W = beta * virial passed as ``uv`` and ``dxdq = sum_i dx/dq_i q_i``.
``<dxdq>`` of every bootstrap replicate is a first-moment reduction of the
``dxdqv`` samples with the replicate's counts -- the same reduce / resample
kernels with order 0 -- instead of a gather by ``sampler.indices``
(reference volume.py:121-126).
"""

from __future__ import annotations

from functools import lru_cache

import torch

from . import engine
from . import moments as cmomy
from . import symbolic as S
from .data import DataCallbackABC, DataValues, xrwrap_xv
from .models import Derivatives, ExtrapModel
from .xrlite import DataArray, as_labelled


class _FirstOrderSeries:
    """series[0], series[1]; anything higher raises like the reference (volume.py:46-54)."""

    def __init__(self, polys):
        self._p = list(polys)

    def __getitem__(self, order):
        if order > 1:
            raise ValueError(
                "Volume derivatives cannot go past 1st order"
                + " and received %i" % order
                + "\n(because would need derivatives of forces)"
            )
        return self._p[order]


def _volume_polys():
    xw0, xw1, w1 = S.xu_raw(0), S.xu_raw(1), S.u_raw(1)
    inv = S.Poly.atom(("volume",), -1) * S.Poly.atom(("ndim",), -1)
    return [xw0, (-(xw0 * w1) + xw1 + S.Poly.atom(("dxdq",))) * inv]


@lru_cache(5)
def factory_derivatives():
    """Derivatives object for the volume expansion."""
    return Derivatives(_FirstOrderSeries(_volume_polys()), args=("W", "xW", "dxdq", "volume", "ndim"))


class VolumeDataCallback(DataCallbackABC):
    """Extra derivative arguments <dxdq>, volume, ndim (reference volume.py:90-134)."""

    _fields = ("volume", "dxdqv", "ndim")

    def __init__(self, volume, dxdqv, ndim=3, _sampler=None, _rep_dim=None):
        if not isinstance(volume, float):
            raise TypeError("volume must be a float")
        if not isinstance(ndim, int):
            raise TypeError("ndim must be an int")
        self.volume, self.dxdqv, self.ndim = volume, dxdqv, ndim
        self._sampler, self._rep_dim = _sampler, _rep_dim
        self._cache = {}

    def check(self, data) -> None:
        pass

    def _dxdq_state(self, rec_dim):
        """comoment state (order 0) of dxdqv: [..., 2, 1]; [.., 1, 0] is the mean."""
        if "st" not in self._cache:
            ones = DataArray(as_labelled(self.dxdqv).isel({d: 0 for d in self.dxdqv.dims if d != rec_dim}).values * 0.0,
                             rec_dim)
            if self._sampler is None:
                st = cmomy.wrap_reduce_vals(self.dxdqv, ones, mom=(1, 0), dim=rec_dim)
            else:
                st = cmomy.wrap_resample_vals(self.dxdqv, ones, mom=(1, 0), sampler=self._sampler, dim=rec_dim,
                                              rep_dim=self._rep_dim)
            self._cache["st"] = st
        return self._cache["st"]

    def dxdq(self, rec_dim):
        """<dxdq> (per replicate after resampling)."""
        return self._dxdq_state(rec_dim).select_moment("xave")

    def resample(self, data, meta_kws, sampler, rep_dim="rep", **kws):
        if not isinstance(data, DataValues):
            raise NotImplementedError("resampling only possible with DataValues style.")
        return VolumeDataCallback(self.volume, self.dxdqv, self.ndim, _sampler=sampler, _rep_dim=rep_dim)

    def derivs_args(self, data, derivs_args):
        return (*tuple(derivs_args), self.dxdq(data.rec_dim), self.volume, self.ndim)

    # device hook used by models.Derivatives
    def device_sources(self, data, src, srcs):
        st = self._dxdq_state(data.rec_dim)
        lead = [self._rep_dim] if (self._sampler is not None and self._rep_dim in st.val_dims) else []
        rest = [d for d in st.val_dims if d not in lead]
        t = st.transpose(*lead, *rest, *st.mom_dims).device_values.contiguous()  # (rep?, val..., 2, 1)
        consts = engine.const_tensor([float(self.volume), float(self.ndim)], torch.float64)
        i_dx, i_c = len(srcs), len(srcs) + 1
        srcs.extend([t, consts])
        s_rep = src.nval * 2 if lead else 0
        return {
            "dxdq": lambda a: (i_dx, 1, s_rep, 2),
            "volume": lambda a: (i_c, 0, 0, 0),
            "ndim": lambda a: (i_c, 1, 0, 0),
        }


def factory_extrapmodel(volume, uv, xv, dxdqv, ndim=3, order=1, alpha_name="volume", rec_dim="rec",
                        val_dims="val", rep_dim="rep", **kws):
    """ExtrapModel for a first-order volume expansion (reference volume.py:137-233).
    ``uv`` is the temperature-scaled virial ``beta * virial``."""
    if order != 1:
        raise ValueError("only order=1 is supported")
    from .data import xrwrap_uv

    uv = xrwrap_uv(uv, rec_dim=rec_dim, rep_dim=rep_dim)
    xv = xrwrap_xv(xv, rec_dim=rec_dim, rep_dim=rep_dim, deriv_dim=None, val_dims=val_dims)
    dxdqv = xrwrap_xv(dxdqv, rec_dim=rec_dim, rep_dim=rep_dim, deriv_dim=None, val_dims=val_dims)
    meta = VolumeDataCallback(volume=float(volume), dxdqv=dxdqv, ndim=ndim)
    data = DataValues.from_vals(uv=uv, xv=xv, order=order, meta=meta, rec_dim=rec_dim, deriv_dim=None, **kws)
    return ExtrapModel(alpha0=volume, data=data, derivatives=factory_derivatives(), order=order, minus_log=False,
                       alpha_name=alpha_name)
