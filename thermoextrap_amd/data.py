"""Data handlers: the thermoextrap.data API on top of the MI355X moment engine.

Mirrors /root/reference/src/thermoextrap/data.py (class and method names,
argument meaning, dims conventions, error behaviour) so that callers -- and the
parity tests, which read like the reference's tests/test_data.py -- can switch
packages without edits.  All moment arithmetic goes through
:mod:`thermoextrap_amd.moments` (the cmomy-API mirror) and therefore through
libtxmom on the GPU.

Conventions (reference data.py:1-12):
  uv, xv : samples of u and x;  u[i] = <u^i>;  xu[i] = <x u^i>;
  with ``deriv_dim``: xu[i, j] = <d^j x/d beta^j  u^i>.
The comoment state ``dxduave`` has trailing dims (xmom=2, umom=order+1).
"""

from __future__ import annotations

import copy as _copy
from collections.abc import Mapping
from typing import Any, Callable

import numpy as np

from . import moments as cmomy
from .moments import MISSING, CentralMomentsData, DeviceDataArray
from .xrlite import DataArray, Dataset, as_dataset, as_labelled, concat, is_dataset, is_labelled

__all__ = [
    "DataCallback", "DataCallbackABC", "DataCentralMoments", "DataCentralMomentsVals", "DataSelector",
    "DataValues", "DataValuesCentral", "factory_data_values", "xrwrap_uv", "xrwrap_xv", "xrwrap_alpha",
]


# ---------------------------------------------------------------------------
# dim labelling (reference core/xrutils.py:55-134)
# ---------------------------------------------------------------------------
def _labelled(x) -> bool:
    return is_labelled(x) or isinstance(x, DeviceDataArray)


def _wrap(x, dims_by_ndim: Mapping[int, list], name, strict):
    if _labelled(x):
        x = x if isinstance(x, DeviceDataArray) else as_labelled(x)
        if strict:
            for d in dims_by_ndim[len(x.dims)]:
                if d not in x.dims:
                    raise ValueError(f"{d} not in dims")
        return x
    x = np.asarray(x)
    return DataArray(x, dims_by_ndim[x.ndim], name=name)


def xrwrap_uv(uv, dims=None, rec_dim="rec", rep_dim="rep", name="u", strict=True):
    """uv[rec] or uv[rep, rec]."""
    dims = dims or {1: [rec_dim], 2: [rep_dim, rec_dim]}
    return _wrap(uv, dims, name, strict)


def xrwrap_xv(xv, dims=None, rec_dim="rec", rep_dim="rep", deriv_dim=None, val_dims="val", name="x", strict=None):
    """xv[rec], xv[rec, val...], xv[rep, rec, val...]; with deriv_dim: xv[rec, deriv(, val...)]."""
    vd = [val_dims] if isinstance(val_dims, str) else list(val_dims)
    if dims is None:
        if deriv_dim is None:
            rec_val = [rec_dim, *vd]
            rep_val = [rep_dim, rec_dim, *vd]
            dims = {1: [rec_dim], len(rec_val): rec_val, len(rep_val): rep_val}
        else:
            rec_val = [rec_dim, deriv_dim, *vd]
            rep_val = [rep_dim, rec_dim, deriv_dim, *vd]
            dims = {2: [rec_dim, deriv_dim], len(rec_val): rec_val, len(rep_val): rep_val}
    return _wrap(xv, dims, name, bool(strict))


def xrwrap_alpha(alpha, dims=None, name="alpha") -> DataArray:
    if is_labelled(alpha):
        return as_labelled(alpha)
    a = np.array(alpha)
    dims = dims or name
    if a.ndim == 0:
        return DataArray(a, (), coords={dims: a}, name=name)
    if a.ndim == 1:
        return DataArray(a, dims, coords={dims: a}, name=name)
    return DataArray(a, dims, name=name)


# ---------------------------------------------------------------------------
# Dataset-valued observables (reference data.py:347-350: ``xv`` may be an xr.Dataset)
# ---------------------------------------------------------------------------
DS_DIM = "_dsvar"  # the column axis of a stacked Dataset


def stack_dataset(ds, rec_dim):
    """All variables of ``ds`` as ONE (rec, column) matrix -- each variable's non-record dims flattened into a block
    of columns -- so that a Dataset costs one launch per kernel instead of one per variable.  Returns the matrix
    (DataArray, or DeviceDataArray when every variable is one) and the layout ``unstack_dataset`` needs."""
    ds = as_dataset(ds)
    if len(ds) == 0:
        raise ValueError("empty Dataset")
    on_device = all(isinstance(v, DeviceDataArray) for v in ds.values())
    blocks, layout, start, nrec = [], [], 0, None
    for name, v in ds.items():
        if rec_dim not in v.dims:
            raise ValueError(f"Dataset variable {name!r} has no {rec_dim!r} dimension")
        ax = v.dims.index(rec_dim)
        other = tuple(d for d in v.dims if d != rec_dim)
        if isinstance(v, DeviceDataArray):
            t = v.tensor.movedim(ax, 0)
            shape = tuple(t.shape[1:])
            blk = t.reshape(t.shape[0], -1)
            coords = {}
            if not on_device:
                blk = blk.cpu().numpy()
        else:
            a = np.moveaxis(np.asarray(v.values, dtype=np.float64), ax, 0)
            shape = a.shape[1:]
            blk = a.reshape(a.shape[0], -1)
            coords = {k: (cd, cv) for k, (cd, cv) in v._coords.items() if cd and all(d in other for d in cd)}
        if nrec is None:
            nrec = blk.shape[0]
        elif blk.shape[0] != nrec:
            raise ValueError("Dataset variables differ in the length of the record dimension")
        layout.append((name, start, start + blk.shape[1], other, tuple(shape), coords))
        start += blk.shape[1]
        blocks.append(blk)
    if on_device:
        import torch

        return DeviceDataArray(torch.cat(blocks, dim=1).contiguous(), (rec_dim, DS_DIM)), tuple(layout)
    return DataArray(np.ascontiguousarray(np.concatenate(blocks, axis=1)), (rec_dim, DS_DIM)), tuple(layout)


def unstack_dataset(arr, layout) -> Dataset:
    """Split the stacked column axis of a result back into the variables of the Dataset it came from."""
    arr = as_labelled(arr)
    ax = arr.dims.index(DS_DIM)
    out = {}
    for name, lo, hi, dims, shape, coords in layout:
        vals = np.take(arr.values, np.arange(lo, hi), axis=ax)
        vals = vals.reshape(arr.shape[:ax] + tuple(shape) + arr.shape[ax + 1:])
        v = DataArray(vals, arr.dims[:ax] + tuple(dims) + arr.dims[ax + 1:], name=name)
        v._inherit({k: c for k, c in arr._coords.items() if DS_DIM not in c[0]})
        v._inherit(coords)
        out[name] = v
    return Dataset(out)


def _concat_rec(a, b, dim):
    """Two labelled sample arrays joined along ``dim`` (host or device resident; the dims of ``a`` rule)."""
    from .moments import DeviceDataArray

    if isinstance(a, DeviceDataArray) or isinstance(b, DeviceDataArray):
        import torch

        from . import engine

        ta = a.tensor if isinstance(a, DeviceDataArray) else engine.to_device(as_labelled(a).values)
        tb = b.tensor if isinstance(b, DeviceDataArray) else engine.to_device(as_labelled(b).values)
        da = tuple(a.dims)
        db = tuple(b.dims)
        if set(da) != set(db):
            raise ValueError(f"dims differ: {da} vs {db}")
        tb = tb.permute([db.index(d) for d in da])
        return DeviceDataArray(torch.cat([ta, tb], dim=da.index(dim)), da)
    if not is_labelled(a) and not is_labelled(b):  # plain 1-D arrays (weights)
        return np.concatenate([np.asarray(a), np.asarray(b)])
    la, lb = as_labelled(a, dims=(dim,)), as_labelled(b, dims=(dim,))
    if set(la.dims) != set(lb.dims):
        raise ValueError(f"dims differ: {la.dims} vs {lb.dims}")
    lb = lb.transpose(*la.dims)
    out = DataArray(np.concatenate([np.asarray(la.values), np.asarray(lb.values)], axis=la.dims.index(dim)), la.dims)
    out._inherit({k: v for k, v in la._coords.items() if dim not in v[0]})
    return out


def _need_dataarray(x, name=None):
    if not _labelled(x):
        raise TypeError(f"type({name})={type(x)} must be a DataArray.")


# ---------------------------------------------------------------------------
# small helpers
# ---------------------------------------------------------------------------
class _Params:
    """new_like / set_params for plain classes (the reference gets these from attrs)."""

    _fields: tuple = ()

    def _asdict(self) -> dict:
        return {f: getattr(self, f) for f in self._fields}

    def new_like(self, **kws):
        d = self._asdict()
        d.update(kws)
        return type(self)(**d)

    def set_params(self, **kws):
        out = _copy.copy(self)
        for k, v in kws.items():
            if k not in self._fields:
                raise ValueError(f"{k} is not a parameter of {type(self).__name__}")
            setattr(out, k, v)
        if hasattr(out, "_cache"):
            out._cache = {}
        return out


def _memo(fn: Callable):
    """memoise a no-argument method/property body in self._cache (honours _use_cache)."""
    key = fn.__name__

    def wrapper(self):
        if getattr(self, "_use_cache", True):
            if key not in self._cache:
                self._cache[key] = fn(self)
            return self._cache[key]
        return fn(self)

    wrapper.__name__ = key
    wrapper.__doc__ = fn.__doc__
    return wrapper


class DataSelector(_Params):
    """Index a labelled array like ``ds[i, j]`` along named dims (reference data.py:91-162)."""

    _fields = ("data", "dims")

    def __init__(self, data, dims):
        if not is_labelled(data):
            raise TypeError("data must be a DataArray")
        data = as_labelled(data)
        dims = (dims,) if isinstance(dims, str) else tuple(dims)
        for d in dims:
            if d not in data.dims:
                raise ValueError(f"{d} not in data.dimensions {data.dims}")
        self.data, self.dims = data, dims

    @classmethod
    def from_defaults(cls, data, *, dims=None, mom_dim="moment", deriv_dim=None):
        if dims is None:
            dims = (mom_dim, deriv_dim) if deriv_dim is not None else (mom_dim,)
        return cls(data=data, dims=dims)

    def __getitem__(self, idx):
        if not isinstance(idx, tuple):
            idx = (idx,)
        if len(idx) != len(self.dims):
            raise ValueError(f"bad idx {idx}, vs dims {self.dims}")
        return self.data.isel(dict(zip(self.dims, idx)), drop=True)

    def __repr__(self):
        return repr(self.data)


class DataCallbackABC(_Params):
    """Hook to adjust a data object (extra derivative arguments, resampling of
    auxiliary samples): reference data.py:165-217."""

    def check(self, data) -> None:
        raise NotImplementedError

    def derivs_args(self, data, *, derivs_args: tuple) -> tuple:
        raise NotImplementedError

    def resample(self, data, *, meta_kws, sampler, **kws):
        raise NotImplementedError

    def reduce(self, data, *, meta_kws, **kws):
        raise NotImplementedError

    def __repr__(self):
        return f"<{type(self).__name__}>"


class DataCallback(DataCallbackABC):
    """Pass-through callback (default)."""

    def check(self, data) -> None:
        pass

    def derivs_args(self, data, *, derivs_args):
        return derivs_args

    def resample(self, data, *, meta_kws, sampler, **kws):
        return self

    def reduce(self, data, *, meta_kws, **kws):
        return self


def _coerce_meta(meta, owner):
    if meta is None:
        meta = DataCallback()
    if not isinstance(meta, DataCallbackABC):
        raise TypeError("meta must be None or subclass of DataCallbackABC")
    meta.check(data=owner)
    return meta


class AbstractData(_Params):
    umom_dim = "umom"
    deriv_dim = None
    x_is_u = False

    @property
    def xalpha(self) -> bool:
        """Whether x depends explicitly on alpha (``deriv_dim`` given)."""
        return self.deriv_dim is not None

    def pipe(self, func, *args, **kwargs):
        return func(self, *args, **kwargs)


# ---------------------------------------------------------------------------
# value-based classes (reference data.py:326-730)
# ---------------------------------------------------------------------------
def build_aves_xu(uv, xv, *, order, dim=MISSING, umom_dim="umom"):
    """(u, xu): raw moments <u^k>, <x u^k>, k <= order (reference data.py:456-493)."""
    _need_dataarray(uv, "uv")
    _need_dataarray(xv, "xv")
    u = cmomy.wrap_reduce_vals(uv, mom=order, dim=dim, mom_dims=umom_dim).rmom()
    xu = cmomy.select_moment(
        cmomy.wrap_reduce_vals(xv, uv, mom=(1, order), dim=dim, mom_dims=("_xmom", umom_dim)).rmom(),
        "xmom_1", mom_ndim=2,
    )
    return u, xu


def build_aves_dxdu(uv, xv, *, order, dim=MISSING, umom_dim="umom"):
    """(xave, duave, dxduave): central forms (reference data.py:496-536)."""
    _need_dataarray(uv, "uv")
    _need_dataarray(xv, "xv")
    duave = cmomy.wrap_reduce_vals(uv, mom=order, dim=dim, mom_dims=umom_dim).cmom()
    c = cmomy.wrap_reduce_vals(xv, uv, mom=(1, order), dim=dim, mom_dims=("_xmom", umom_dim))
    xave = c.select_moment("xave")
    dxduave = cmomy.select_moment(c.cmom(), "xmom_1", mom_ndim=2)
    return xave, duave, dxduave


def _xu_to_u(xu: DataArray, dim="umom") -> DataArray:
    """x == u: shift umom by one and prepend the zeroth moment 1 (reference data.py:539-546)."""
    lead = [d for d in xu.dims if d != dim]
    v = xu.transpose(*lead, dim).values
    out = np.empty(v.shape[:-1] + (v.shape[-1] + 1,))
    out[..., 0] = 1.0
    out[..., 1:] = v
    return DataArray(out, (*lead, dim)).transpose(*xu.dims)


class DataValuesBase(AbstractData):
    """Holds raw samples; moments are computed on demand (reference data.py:326-449).

    ``resample`` follows the reference's gather semantics
    (``uv.isel(rec=indices)``); on this engine the gathered (rep, rec) copies
    are never materialised unless ``.uv``/``.xv`` of the resampled object are
    read -- the moments of replicate r are the freq-weighted comoments, which is
    the identity the reference asserts in tests/test_data.py:94-112.
    """

    _CENTRAL = False
    _fields = ("uv", "xv", "order", "rec_dim", "umom_dim", "deriv_dim", "meta", "x_is_u")

    def __init__(self, uv, xv, order, *, rec_dim="rec", umom_dim="umom", deriv_dim=None, meta=None, x_is_u=False,
                 _resampled=None):
        if not _labelled(uv) or (hasattr(uv, "dims") and False):
            raise TypeError("uv must be a DataArray")
        if not _labelled(xv):
            raise TypeError("xv must be a DataArray")
        self._uv, self._xv = uv, xv
        self.order = order
        self.rec_dim, self.umom_dim, self.deriv_dim, self.x_is_u = rec_dim, umom_dim, deriv_dim, x_is_u
        self._cache: dict[str, Any] = {}
        self._resampled = _resampled  # (base_uv, base_xv, sampler, rep_dim) when made by .resample
        self.meta = _coerce_meta(meta, self)

    @classmethod
    def from_vals(cls, uv, xv, *, order, rec_dim="rec", umom_dim="umom", deriv_dim=None, meta=None, x_is_u=False):
        return cls(uv=uv, xv=uv if xv is None else xv, order=order, rec_dim=rec_dim, umom_dim=umom_dim,
                   deriv_dim=deriv_dim, meta=meta, x_is_u=x_is_u)

    @property
    def central(self) -> bool:
        return self._CENTRAL

    # gathered views are built lazily for resampled objects
    @property
    def uv(self):
        if self._uv is None:
            buv, _, sampler, rep_dim = self._resampled
            idx = DataArray(sampler.indices, (rep_dim, self.rec_dim))
            self._uv = as_labelled(buv).isel({self.rec_dim: idx})
        return self._uv

    @property
    def xv(self):
        if self._xv is None:
            buv, bxv, sampler, rep_dim = self._resampled
            if self.x_is_u:
                self._xv = self.uv
            else:
                idx = DataArray(sampler.indices, (rep_dim, self.rec_dim))
                self._xv = as_labelled(bxv).isel({self.rec_dim: idx})
        return self._xv

    def __len__(self):
        src = self._resampled[0] if self._resampled is not None else self._uv
        return int(src.sizes[self.rec_dim])

    def resample(self, sampler, *, rep_dim="rep", meta_kws=None):
        if self._resampled is not None:
            raise NotImplementedError("resampling an already resampled value object")
        sampler = cmomy.factory_sampler(sampler, data=self._xv, dim=self.rec_dim, rep_dim=rep_dim)
        if sampler.ndat != len(self) or (sampler._indices is not None and sampler._indices.shape[1] != len(self)):
            n = sampler._indices.shape[1] if sampler._indices is not None else sampler.ndat
            raise ValueError(f"indices.sizes[{self.rec_dim}]={n} must equal len(self)={len(self)}")
        meta = self.meta.resample(data=self, meta_kws={} if meta_kws is None else meta_kws, sampler=sampler,
                                  rep_dim=rep_dim)
        out = type(self)(uv=self._uv, xv=self._xv, order=self.order, rec_dim=self.rec_dim, umom_dim=self.umom_dim,
                         deriv_dim=self.deriv_dim, meta=meta, x_is_u=self.x_is_u,
                         _resampled=(self._uv, self._xv, sampler, rep_dim))
        out._uv = out._xv = None
        return out

    # comoment state of the (possibly resampled) samples
    def _state(self, x_as_u=False) -> CentralMomentsData:
        key = "_state_u" if x_as_u else "_state_x"
        if key not in self._cache:
            if self._resampled is None:
                xv = self._uv if x_as_u else self._xv
                st = cmomy.wrap_reduce_vals(xv, self._uv, mom=(1, self.order), dim=self.rec_dim,
                                            mom_dims=("_xmom", self.umom_dim))
            else:
                buv, bxv, sampler, rep_dim = self._resampled
                xv = buv if x_as_u else bxv
                st = cmomy.wrap_resample_vals(xv, buv, mom=(1, self.order), sampler=sampler, dim=self.rec_dim,
                                              rep_dim=rep_dim, mom_dims=("_xmom", self.umom_dim))
            self._cache[key] = st
        return self._cache[key]


class DataValues(DataValuesBase):
    """Raw-moment view of uv/xv samples (reference data.py:549-593)."""

    _CENTRAL = False

    @property
    def xu(self) -> DataArray:
        """<x u^n>"""
        if "xu" not in self._cache:
            self._cache["xu"] = cmomy.select_moment(self._state().rmom(), "xmom_1", mom_ndim=2)
        return self._cache["xu"]

    @property
    def u(self) -> DataArray:
        """<u^n>"""
        if "u" not in self._cache:
            if self.x_is_u:
                self._cache["u"] = _xu_to_u(self.xu, self.umom_dim)
            else:
                # moments of u alone: the x == u comoment state's first row
                self._cache["u"] = cmomy.select_moment(self._state(x_as_u=True).rmom(), "xmom_0", mom_ndim=2)
        return self._cache["u"]

    @property
    def u_selector(self):
        return DataSelector.from_defaults(self.u, deriv_dim=None, mom_dim=self.umom_dim)

    @property
    def xu_selector(self):
        return DataSelector.from_defaults(self.xu, deriv_dim=self.deriv_dim, mom_dim=self.umom_dim)

    @property
    def derivs_args(self) -> tuple:
        out = (self.u_selector,) if self.x_is_u else (self.u_selector, self.xu_selector)
        return self.meta.derivs_args(data=self, derivs_args=out)

    # layout hook used by Derivatives to evaluate on the device
    def _derivs_source(self):
        return _source_from_state(self._state(), central=False, x_is_u=self.x_is_u, deriv_dim=self.deriv_dim,
                                  lead_dim=self._lead_dim())

    def _lead_dim(self):
        return self._resampled[3] if self._resampled is not None else None


class DataValuesCentral(DataValuesBase):
    """Central-moment view of uv/xv samples (reference data.py:596-656)."""

    _CENTRAL = True

    @property
    def xave(self) -> DataArray:
        if "xave" not in self._cache:
            self._cache["xave"] = self._state().select_moment("xave")
        return self._cache["xave"]

    @property
    def dxdu(self) -> DataArray:
        if "dxdu" not in self._cache:
            self._cache["dxdu"] = cmomy.select_moment(self._state().cmom(), "xmom_1", mom_ndim=2)
        return self._cache["dxdu"]

    @property
    def du(self) -> DataArray:
        if "du" not in self._cache:
            if self.x_is_u:
                self._cache["du"] = _xu_to_u(self.dxdu, dim=self.umom_dim)
            else:
                self._cache["du"] = cmomy.select_moment(self._state(x_as_u=True).cmom(), "xmom_0", mom_ndim=2)
        return self._cache["du"]

    @property
    def du_selector(self):
        return DataSelector.from_defaults(self.du, deriv_dim=None, mom_dim=self.umom_dim)

    @property
    def dxdu_selector(self):
        return DataSelector.from_defaults(self.dxdu, deriv_dim=self.deriv_dim, mom_dim=self.umom_dim)

    @property
    def xave_selector(self):
        if self.deriv_dim is None:
            return self.xave
        return DataSelector.from_defaults(self.xave, dims=[self.deriv_dim])

    @property
    def derivs_args(self) -> tuple:
        out = ((self.xave_selector, self.du_selector) if self.x_is_u
               else (self.xave_selector, self.du_selector, self.dxdu_selector))
        return self.meta.derivs_args(data=self, derivs_args=out)

    def _derivs_source(self):
        return _source_from_state(self._state(), central=True, x_is_u=self.x_is_u, deriv_dim=self.deriv_dim,
                                  lead_dim=self._lead_dim())

    def _lead_dim(self):
        return self._resampled[3] if self._resampled is not None else None


def factory_data_values(order, uv, xv, central=False, xalpha=False, rec_dim="rec", umom_dim="umom", val_dims="val",
                        rep_dim="rep", deriv_dim=None, x_is_u=False, **kws):
    """DataValues / DataValuesCentral from arrays (reference data.py:659-730)."""
    cls = DataValuesCentral if central else DataValues
    if xalpha and deriv_dim is None:
        raise ValueError("if xalpha, must pass string name of derivative")
    uv = xrwrap_uv(uv, rec_dim=rec_dim, rep_dim=rep_dim)
    if xv is not None:
        xv = xrwrap_xv(xv, rec_dim=rec_dim, rep_dim=rep_dim, deriv_dim=deriv_dim, val_dims=val_dims)
    return cls.from_vals(uv=uv, xv=xv, order=order, rec_dim=rec_dim, umom_dim=umom_dim, deriv_dim=deriv_dim,
                         x_is_u=x_is_u, **kws)


# ---------------------------------------------------------------------------
# device layout description consumed by models.Derivatives
# ---------------------------------------------------------------------------
class DerivSource:
    """Where the scalars of the derivative formulas live on the device.

    ``tensor`` is contiguous with dims (lead?, deriv?, val..., 2, K) holding either
    the cmomy state (central) or its raw-moment conversion.  ``resolve(kind, n, d)``
    maps a symbol occurrence to (offset, stride_rep, stride_val)."""

    def __init__(self, tensor, *, central, x_is_u, nrep, ndrv, nval, out_dims, out_shape, coords):
        self.tensor, self.central, self.x_is_u = tensor, central, x_is_u
        self.nrep, self.ndrv, self.nval = nrep, ndrv, nval
        self.out_dims, self.out_shape, self.coords = out_dims, out_shape, coords
        self.K = tensor.shape[-1]

    def resolve(self, kind: str, n: int = 0, d: int = 0):
        K, V, D = self.K, self.nval, self.ndrv
        s_rep, s_val = D * V * 2 * K, 2 * K
        if d >= D:
            raise ValueError(f"derivative index {d} not available (deriv dim has {D} entries)")
        base = d * V * 2 * K
        if kind in ("du", "u"):
            # moments of u: first row at deriv = 0; x_is_u supplies order+1 from the x row
            if n < K:
                return (0 * K + n, s_rep, s_val)
            if self.x_is_u and n == K:
                return (1 * K + (K - 1), s_rep, s_val)
            raise ValueError(f"moment {kind}[{n}] exceeds the stored order {K - 1}")
        if kind in ("dxdu", "xu"):
            if n >= K:
                raise ValueError(f"moment {kind}[{n}] exceeds the stored order {K - 1}")
            return (base + K + n, s_rep, s_val)
        if kind == "x1":
            return (base + K, s_rep, s_val)
        if kind == "umean":
            return (1, s_rep, s_val)
        raise ValueError(f"unknown symbol kind {kind}")


def _source_from_state(state: CentralMomentsData, *, central, x_is_u, deriv_dim, lead_dim) -> DerivSource:
    dims = state.dims
    vdims = list(state.val_dims)
    lead = lead_dim if (lead_dim is not None and lead_dim in vdims) else None
    rest = [d for d in vdims if d not in (lead, deriv_dim)]
    order = ([lead] if lead else []) + ([deriv_dim] if deriv_dim in vdims else []) + rest + list(state.mom_dims)
    st = state if tuple(order) == dims else state.transpose(*order)
    t = st.device_values if central else st.rmom_device()
    t = t.contiguous()
    sz = st.sizes
    nrep = sz[lead] if lead else 1
    ndrv = sz[deriv_dim] if deriv_dim in vdims else 1
    nval = int(np.prod([sz[d] for d in rest])) if rest else 1
    out_dims = ([lead] if lead else []) + rest
    out_shape = [sz[d] for d in out_dims]
    coords = {k: v for k, v in st._coords.items() if all(d in out_dims for d in v[0])}
    return DerivSource(t, central=central, x_is_u=x_is_u, nrep=nrep, ndrv=ndrv, nval=nval, out_dims=out_dims,
                       out_shape=out_shape, coords=coords)


# ---------------------------------------------------------------------------
# comoment-state based classes (reference data.py:791-1813)
# ---------------------------------------------------------------------------
class DataCentralMomentsBase(AbstractData):
    _fields = ("dxduave", "xmom_dim", "umom_dim", "rec_dim", "deriv_dim", "central", "meta", "x_is_u", "use_cache",
               "ds_layout")

    def _init_base(self, dxduave, *, xmom_dim, umom_dim, rec_dim, deriv_dim, central, meta, x_is_u, use_cache,
                   ds_layout=None):
        if not isinstance(dxduave, CentralMomentsData):
            raise TypeError("dxduave must be a CentralMomentsData")
        self.dxduave = dxduave
        self.xmom_dim, self.umom_dim, self.rec_dim, self.deriv_dim = xmom_dim, umom_dim, rec_dim, deriv_dim
        self.central, self.x_is_u = bool(central), bool(x_is_u)
        self._use_cache = use_cache
        self._cache: dict[str, Any] = {}
        self.meta = _coerce_meta(meta, self)
        # set when ``xv`` was a Dataset: its variables are column blocks of one matrix (stack_dataset); results with
        # the stacked axis are split back by ``as_dataset_result``
        self.ds_layout = ds_layout

    @property
    def use_cache(self):
        return self._use_cache

    def as_dataset_result(self, arr):
        """A result array of a Dataset-valued ``xv`` as a Dataset again (no-op for DataArray-valued data)."""
        if self.ds_layout is None or not is_labelled(arr) or DS_DIM not in as_labelled(arr).dims:
            return arr
        return unstack_dataset(arr, self.ds_layout)

    @property
    def order(self) -> int:
        return self.dxduave.sizes[self.umom_dim] - 1

    @property
    def values(self) -> DataArray:
        """The ``[..., xmom, umom]`` state array (``cmomy.CentralMomentsData.obj``)."""
        return self.dxduave.obj

    @_memo
    def rmom(self) -> DataArray:
        return self.dxduave.rmom()

    @_memo
    def cmom(self) -> DataArray:
        return self.dxduave.cmom()

    def _first_deriv(self, out):
        return out.sel({self.deriv_dim: 0}, drop=True) if self.xalpha else out

    @property
    @_memo
    def xu(self) -> DataArray:
        return cmomy.select_moment(self.rmom(), "xmom_1", mom_ndim=2, mom_dims=self.dxduave.mom_dims)

    @property
    @_memo
    def u(self) -> DataArray:
        if self.x_is_u:
            return cmomy.convert.comoments_to_moments(self.rmom(), mom_dims=self.dxduave.mom_dims,
                                                      mom_dims_out=self.umom_dim)
        return self._first_deriv(
            cmomy.select_moment(self.rmom(), "xmom_0", mom_ndim=2, mom_dims=self.dxduave.mom_dims))

    @property
    @_memo
    def xave(self) -> DataArray:
        return self.dxduave.select_moment("xave")

    @property
    @_memo
    def dxdu(self) -> DataArray:
        return cmomy.select_moment(self.cmom(), "xmom_1", mom_ndim=2, mom_dims=self.dxduave.mom_dims)

    @property
    @_memo
    def du(self) -> DataArray:
        if self.x_is_u:
            return cmomy.convert.comoments_to_moments(self.cmom(), mom_dims=self.dxduave.mom_dims,
                                                      mom_dims_out=self.umom_dim)
        return self._first_deriv(
            cmomy.select_moment(self.cmom(), "xmom_0", mom_ndim=2, mom_dims=self.dxduave.mom_dims))

    @property
    def u_selector(self):
        return DataSelector.from_defaults(self.u, deriv_dim=None, mom_dim=self.umom_dim)

    @property
    def xu_selector(self):
        return DataSelector.from_defaults(self.xu, deriv_dim=self.deriv_dim, mom_dim=self.umom_dim)

    @property
    def xave_selector(self):
        if self.deriv_dim is None:
            return self.xave
        return DataSelector(self.xave, dims=[self.deriv_dim])

    @property
    def du_selector(self):
        return DataSelector.from_defaults(self.du, deriv_dim=None, mom_dim=self.umom_dim)

    @property
    def dxdu_selector(self):
        return DataSelector.from_defaults(self.dxdu, deriv_dim=self.deriv_dim, mom_dim=self.umom_dim)

    @property
    def derivs_args(self) -> tuple:
        """Arguments of the lambdified derivative functions (reference data.py:944-962)."""
        if not self.x_is_u:
            out = ((self.xave_selector, self.du_selector, self.dxdu_selector) if self.central
                   else (self.u_selector, self.xu_selector))
        elif self.central:
            out = (self.xave_selector, self.du_selector)
        else:
            out = (self.u_selector,)
        return self.meta.derivs_args(data=self, derivs_args=out)

    def _derivs_source(self) -> DerivSource:
        lead = self.rec_dim if self.rec_dim in self.dxduave.val_dims else None
        return _source_from_state(self.dxduave, central=self.central, x_is_u=self.x_is_u, deriv_dim=self.deriv_dim,
                                  lead_dim=lead)


class DataCentralMoments(DataCentralMomentsBase):
    """Comoment states (possibly several records of them) -- reference data.py:965-1618."""

    def __init__(self, dxduave, *, xmom_dim="xmom", umom_dim="umom", rec_dim="rec", deriv_dim=None, central=False,
                 meta=None, x_is_u=False, use_cache=True, ds_layout=None):
        self._init_base(dxduave, xmom_dim=xmom_dim, umom_dim=umom_dim, rec_dim=rec_dim, deriv_dim=deriv_dim,
                        central=central, meta=meta, x_is_u=x_is_u, use_cache=use_cache, ds_layout=ds_layout)

    def __len__(self):
        return self.values.sizes[self.rec_dim]

    def reduce(self, dim=MISSING, axis=MISSING, meta_kws=None, **kwargs):
        """Merge the records along ``dim`` (default ``rec_dim``)."""
        if dim is MISSING and axis is MISSING:
            dim = self.rec_dim
        kws = dict(dim=dim, axis=axis, **kwargs)
        return self.new_like(dxduave=self.dxduave.reduce(**kws),
                             meta=self.meta.reduce(data=self, meta_kws=meta_kws, **kws))

    def push_vals(self, xv, uv, weight=None, dim=MISSING, axis=MISSING):
        """Streaming accumulation (north_star; cmomy ``push_vals``, which thermoextrap itself never calls -- SURVEY 0.7):
        a new object whose state is this one's merged with the comoments of the chunk ``xv[rec, val...]``, ``uv[rec]``
        (``weight[rec]``) -- one reduction of the chunk and one merge kernel, the old samples are not needed.  The state
        must not carry a record dimension (reduce it first); ``DataCentralMoments.from_vals`` of a first chunk, then
        ``push_vals`` of the others, equals ``from_vals`` of all samples to rounding."""
        if self.x_is_u:
            raise NotImplementedError("push_vals with x_is_u")
        if dim is MISSING and axis is MISSING:
            dim = self.rec_dim
        if self.rec_dim in self.dxduave.val_dims:
            raise ValueError(f"push_vals needs a state without the record dimension {self.rec_dim!r}: reduce() it first")
        st = cmomy.CentralMomentsData(self.dxduave.device_values.clone(), mom_ndim=2, dims=self.dxduave.dims)
        st._coords = dict(self.dxduave._coords)
        st.push_vals(xv, uv, weight=weight, dim=dim, axis=axis)
        return self.new_like(dxduave=st)

    def resample(self, sampler, dim=MISSING, axis=MISSING, rep_dim="rep", parallel=None, meta_kws=None, **kwargs):
        """Block bootstrap of the records (reference data.py:1000-1055)."""
        if dim is MISSING and axis is MISSING:
            dim = self.rec_dim
        sampler = cmomy.factory_sampler(sampler, data=self.dxduave, dim=dim, axis=axis,
                                        mom_ndim=self.dxduave.mom_ndim, mom_dims=self.dxduave.mom_dims,
                                        rep_dim=rep_dim, parallel=parallel)
        kws = dict(sampler=sampler, dim=dim, axis=axis, rep_dim=rep_dim, parallel=parallel, **kwargs)
        dxdu_new = self.dxduave.resample_and_reduce(**kws).transpose(rep_dim, ...)
        meta = self.meta.resample(data=self, meta_kws=meta_kws, **kws)
        return self.new_like(dxduave=dxdu_new, rec_dim=rep_dim, meta=meta)

    # ---- constructors -----------------------------------------------------
    @classmethod
    def from_raw(cls, raw, rec_dim="rec", xmom_dim="xmom", umom_dim="umom", deriv_dim=None, central=False,
                 x_is_u=False, meta=None, **kwargs):
        """From raw moments ``raw[..., i, j] = <x^i u^j>`` with ``raw[..., 0, 0]`` the weight."""
        if x_is_u:
            data = cmomy.convert.moments_type(raw, mom_ndim=1, mom_dims=umom_dim, to="central", **kwargs)
        else:
            data = cmomy.convert.moments_type(raw, mom_ndim=2, mom_dims=(xmom_dim, umom_dim), to="central", **kwargs)
        return cls.from_data(data, rec_dim=rec_dim, xmom_dim=xmom_dim, umom_dim=umom_dim, deriv_dim=deriv_dim,
                             central=central, meta=meta, x_is_u=x_is_u)

    @classmethod
    def from_vals(cls, uv, xv, order, xmom_dim="xmom", umom_dim="umom", rec_dim="rec", deriv_dim=None,
                  central=False, weight=None, axis=MISSING, dim=MISSING, meta=None, x_is_u=False, **kwargs):
        """From unaveraged samples, reduced along ``dim``/``axis`` (default axis 0)."""
        _need_dataarray(uv)
        if axis is MISSING and dim is MISSING:
            axis = 0
        if xv is None or x_is_u:
            dxduave = cmomy.wrap_reduce_vals(uv, weight=weight, axis=axis, dim=dim, mom=order + 1,
                                             mom_dims=umom_dim, **kwargs).moments_to_comoments(
                mom_dims_out=(xmom_dim, umom_dim), mom=(1, order))
        else:
            ds_layout = None
            if is_dataset(xv):  # the variables become column blocks of one matrix: one reduction for all of them
                red = dim if dim is not MISSING else uv.dims[axis]
                xv, ds_layout = stack_dataset(xv, red)
                dim, axis = red, MISSING
            _need_dataarray(xv)
            dxduave = cmomy.wrap_reduce_vals(xv, uv, weight=weight, axis=axis, dim=dim, mom=(1, order),
                                             mom_dims=(xmom_dim, umom_dim), **kwargs)
            return cls(dxduave=dxduave, xmom_dim=xmom_dim, umom_dim=umom_dim, rec_dim=rec_dim, deriv_dim=deriv_dim,
                       central=central, meta=meta, x_is_u=x_is_u, ds_layout=ds_layout)
        return cls(dxduave=dxduave, xmom_dim=xmom_dim, umom_dim=umom_dim, rec_dim=rec_dim, deriv_dim=deriv_dim,
                   central=central, meta=meta, x_is_u=x_is_u)

    @classmethod
    def from_data(cls, data, rec_dim="rec", xmom_dim="xmom", umom_dim="umom", deriv_dim=None, central=False,
                  meta=None, x_is_u=False, **kwargs):
        """From a state array (layout in the module docstring)."""
        _need_dataarray(data)
        if x_is_u:
            dxduave = cmomy.wrap(data, mom_ndim=1, mom_dims=umom_dim, **kwargs).moments_to_comoments(
                mom_dims_out=(xmom_dim, umom_dim), mom=(1, -1))
        else:
            dxduave = cmomy.wrap(data, mom_ndim=2, mom_dims=(xmom_dim, umom_dim), **kwargs)
        return cls(dxduave=dxduave, xmom_dim=xmom_dim, umom_dim=umom_dim, rec_dim=rec_dim, deriv_dim=deriv_dim,
                   central=central, meta=meta, x_is_u=x_is_u)

    @classmethod
    def from_resample_vals(cls, xv, uv, order, sampler, weight=None, axis=MISSING, dim=MISSING, xmom_dim="xmom",
                           umom_dim="umom", rep_dim="rep", deriv_dim=None, central=False, meta=None, meta_kws=None,
                           x_is_u=False, parallel=None, **kwargs):
        """One-shot bootstrap constructor (reference data.py:1285-1392)."""
        if xv is None or x_is_u:
            xv = uv
        _need_dataarray(xv)
        _need_dataarray(uv)
        if axis is MISSING and dim is MISSING:
            axis = 0
        mom_dims = (xmom_dim, umom_dim)
        sampler = cmomy.factory_sampler(sampler, data=xv, dim=dim, axis=axis, mom_dims=mom_dims, rep_dim=rep_dim,
                                        parallel=parallel)
        dxduave = cmomy.wrap_resample_vals(xv, uv, weight=weight, sampler=sampler, mom=(1, order), axis=axis,
                                           dim=dim, mom_dims=mom_dims, rep_dim=rep_dim, parallel=parallel, **kwargs)
        out = cls(dxduave=dxduave, xmom_dim=xmom_dim, umom_dim=umom_dim, rec_dim=rep_dim, deriv_dim=deriv_dim,
                  central=central, meta=meta, x_is_u=x_is_u)
        return out.set_params(meta=out.meta.resample(
            data=out, meta_kws=meta_kws, sampler=sampler, weight=weight, mom=(1, order), axis=axis, dim=dim,
            mom_dims=mom_dims, rep_dim=rep_dim, **kwargs))

    @classmethod
    def from_ave_raw(cls, u, xu, weight=None, rec_dim="rec", xmom_dim="xmom", umom_dim="umom", deriv_dim=None,
                     central=False, meta=None, x_is_u=False):
        """From pre-averaged raw moments ``u[n] = <u^n>``, ``xu[n] = <x u^n>`` (reference data.py:1394-1473)."""
        _need_dataarray(u)
        u = as_labelled(u)
        if xu is None or x_is_u:
            raw = u.copy()
            if weight is not None:
                raw = cmomy.assign_moment(raw, weight=weight, mom_dims=umom_dim, copy=False)
            raw = raw.transpose(..., umom_dim)
        else:
            _need_dataarray(xu)
            raw = concat((u, as_labelled(xu)), dim=xmom_dim)
            if weight is not None:
                raw = cmomy.assign_moment(raw, weight=weight, mom_dims=(xmom_dim, umom_dim), copy=False)
            raw = raw.transpose(..., xmom_dim, umom_dim)
        return cls.from_raw(raw=raw, xmom_dim=xmom_dim, umom_dim=umom_dim, deriv_dim=deriv_dim, rec_dim=rec_dim,
                            central=central, meta=meta, x_is_u=x_is_u)

    @classmethod
    def from_ave_central(cls, du, dxdu, weight=None, xave=None, uave=None, axis=-1, umom_axis=None,
                         xumom_axis=None, rec_dim="rec", xmom_dim="xmom", umom_dim="umom", deriv_dim=None,
                         central=False, dtype=None, dims=None, attrs=None, coords=None, name=None, meta=None,
                         x_is_u=False):
        """From pre-averaged central moments (reference data.py:1475-1618):
        ``du[0] = 1|weight, du[1] = <u>|uave, du[n] = <du^n>``;
        ``dxdu[0] = <x>|xave, dxdu[n] = <dx du^n>``."""
        if dxdu is None or x_is_u:
            d = as_labelled(du)
            n = d.sizes[umom_dim]
            dxdu = d.isel({umom_dim: slice(1, None)})
            du = d.isel({umom_dim: slice(None, n - 1)})
        if (xave is None or x_is_u) and uave is not None:
            xave = uave

        def put(arr, sel, val):
            if val is None:
                return
            v = as_labelled(val) if is_labelled(val) else val
            tgt = arr.values[sel]
            if isinstance(v, DataArray):
                lead = [d for d in arr.dims if d not in (xmom_dim, umom_dim)]
                v = v.transpose(*[d for d in lead if d in v.dims]).values
            arr.values[sel] = np.broadcast_to(np.asarray(v), np.shape(tgt))

        if is_labelled(dxdu):
            data = concat((as_labelled(du), as_labelled(dxdu)), dim=xmom_dim).transpose(..., xmom_dim, umom_dim).copy()
            put(data, (Ellipsis, 0, 0), weight)
            put(data, (Ellipsis, 1, 0), xave)
            put(data, (Ellipsis, 0, 1), uave)
        else:
            axis = -1 if axis is None else axis
            du_ = np.swapaxes(np.asarray(du), axis if umom_axis is None else umom_axis, -1)
            dxdu_ = np.swapaxes(np.asarray(dxdu), axis if xumom_axis is None else xumom_axis, -1)
            K = min(du_.shape[-1], dxdu_.shape[-1])
            vals = np.empty(dxdu_.shape[:-1] + (2, K), dtype=dtype or dxdu_.dtype)
            vals[..., 0, :] = du_[..., :K]
            vals[..., 1, :] = dxdu_[..., :K]
            if weight is not None:
                vals[..., 0, 0] = weight
            if xave is not None:
                vals[..., 1, 0] = xave
            if uave is not None:
                vals[..., 0, 1] = uave
            lead = list(dims) if dims is not None else [f"dim_{i}" for i in range(vals.ndim - 2)]
            data = DataArray(vals, (*lead, xmom_dim, umom_dim), coords=coords, name=name, attrs=attrs)
        dxduave = CentralMomentsData(data, mom_ndim=2, mom_dims=(xmom_dim, umom_dim))
        return cls(dxduave=dxduave, xmom_dim=xmom_dim, umom_dim=umom_dim, rec_dim=rec_dim, deriv_dim=deriv_dim,
                   central=central, meta=meta, x_is_u=x_is_u)


class DataCentralMomentsVals(DataCentralMomentsBase):
    """Keeps the samples and their reduced comoment state; ``resample`` bootstraps
    the samples (reference data.py:1643-1813).  The state is reduced eagerly at
    construction, as in the reference (``_convert_dxduave``, data.py:1621-1640)."""

    _fields = ("uv", "xv", "order", "weight", "from_vals_kws", "dxduave", "xmom_dim", "umom_dim", "rec_dim",
               "deriv_dim", "central", "meta", "x_is_u", "use_cache", "ds_layout")

    def __init__(self, uv, xv, *, order=None, weight=None, from_vals_kws=None, dxduave=None, xmom_dim="xmom",
                 umom_dim="umom", rec_dim="rec", deriv_dim=None, central=False, meta=None, x_is_u=False,
                 use_cache=True, ds_layout=None):
        if not _labelled(uv):
            raise TypeError("uv must be a DataArray")
        if is_dataset(xv):  # reference data.py:347-350; the variables become column blocks of one sample matrix
            xv, ds_layout = stack_dataset(xv, rec_dim)
        if not _labelled(xv):
            raise TypeError("xv must be a DataArray or Dataset")
        if order is not None and not isinstance(order, (int, np.integer)):
            raise TypeError("order must be an int")
        self.uv, self.xv, self.order_, self.weight = uv, xv, order, weight
        self.from_vals_kws = dict(from_vals_kws or {})
        if dxduave is None:
            if order is None or order <= 0:
                raise ValueError("must pass order if calculating dxduave")
            dxduave = cmomy.wrap_reduce_vals(xv, uv, weight=weight, dim=rec_dim, mom=(1, order),
                                             mom_dims=(xmom_dim, umom_dim), **self.from_vals_kws)
        self._init_base(dxduave, xmom_dim=xmom_dim, umom_dim=umom_dim, rec_dim=rec_dim, deriv_dim=deriv_dim,
                        central=central, meta=meta, x_is_u=x_is_u, use_cache=use_cache, ds_layout=ds_layout)

    # `order` is a constructor field here but derived from the state elsewhere
    @property
    def order(self) -> int:
        return self.dxduave.sizes[self.umom_dim] - 1

    def _asdict(self):
        d = super()._asdict()
        d["order"] = self.order_
        return d

    @classmethod
    def from_vals(cls, xv, uv, order, weight=None, rec_dim="rec", umom_dim="umom", xmom_dim="xmom", deriv_dim=None,
                  central=False, from_vals_kws=None, meta=None, x_is_u=False):
        return cls(uv=uv, xv=uv if xv is None else xv, order=order, weight=weight, rec_dim=rec_dim,
                   umom_dim=umom_dim, xmom_dim=xmom_dim, deriv_dim=deriv_dim, central=central,
                   from_vals_kws=from_vals_kws, meta=meta, x_is_u=x_is_u)

    def __len__(self):
        return int(self.uv.sizes[self.rec_dim])

    def push_vals(self, xv, uv, weight=None):
        """Append a chunk of samples (``xv``, ``uv`` along ``rec_dim``; ``weight`` iff this object has weights): the new
        object holds the concatenated samples and the MERGED state -- the chunk is reduced and merged into the old state
        (txm_push_vals), the old samples are not read again (cmomy ``push_vals``; north_star's streaming accumulation)."""
        if self.x_is_u:
            raise NotImplementedError("push_vals with x_is_u")
        if (weight is None) != (self.weight is None):
            raise ValueError("give a weight for the chunk exactly when the data object has weights")
        if self.rec_dim in self.dxduave.val_dims:
            raise ValueError("push_vals on a resampled / record-carrying state")
        st = cmomy.CentralMomentsData(self.dxduave.device_values.clone(), mom_ndim=2, dims=self.dxduave.dims)
        st._coords = dict(self.dxduave._coords)
        st.push_vals(xv, uv, weight=weight, dim=self.rec_dim)
        cat = lambda a, b: _concat_rec(a, b, self.rec_dim)  # noqa: E731
        return self.new_like(uv=cat(self.uv, uv), xv=cat(self.xv, xv), weight=None if weight is None else cat(self.weight, weight),
                             dxduave=st)

    def resample(self, sampler, dim=MISSING, axis=MISSING, rep_dim="rep", parallel=None, meta_kws=None, **kwargs):
        """Sample-level bootstrap: draws the sampler, then
        ``wrap_resample_vals(xv, uv, weight, mom=(1, order), sampler)``."""
        if dim is MISSING and axis is MISSING:
            dim = self.rec_dim
        sampler = cmomy.factory_sampler(sampler, data=self.xv, dim=dim, axis=axis, rep_dim=rep_dim, parallel=parallel)
        kws = {"sampler": sampler, "parallel": parallel, "axis": axis, "dim": dim, "rep_dim": rep_dim, **kwargs}
        meta = self.meta.resample(data=self, meta_kws=meta_kws, **kws)
        # the int8 bootstrap path's pre-pass (window scale table, guard flags) depends on the samples only: kept
        # with this object's other cached quantities (reference: per-object cache, data.py:285, 844-942), so a
        # bootstrap loop on one data object runs it once; new_like() starts from an empty cache
        prep = self._cache.get("resample_prep")
        if prep is None:
            from . import engine

            prep = self._cache["resample_prep"] = engine.ResamplePrep()
        dxduave = cmomy.wrap_resample_vals(self.xv, self.uv, weight=self.weight, mom=(1, self.order),
                                           mom_dims=(self.xmom_dim, self.umom_dim), _prep=prep, **kws)
        dxduave = dxduave.transpose(rep_dim, ...)
        return self.new_like(dxduave=dxduave, rec_dim=rep_dim, meta=meta)
