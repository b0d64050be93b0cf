"""A small labelled-array type with the slice of the xarray.DataArray interface
that the reference's hot path relies on (dims, isel/sel, transpose, reductions
by dim name, broadcasting arithmetic by dim name, concat).

xarray itself is not installable in the build image nor on the GPU box, so the
drop-in API (DataCentralMoments, ExtrapModel, ...) returns these objects.  When
xarray *is* importable its DataArrays are accepted wherever a labelled array is
expected (`as_labelled`) and results convert back with `.to_xarray()`.

Only host (numpy) values live here; device residency is the business of
moments.py / engine.py.
"""

from __future__ import annotations

from collections.abc import Hashable, Mapping, Sequence
from typing import Any

import numpy as np

try:  # optional
    import xarray as _xr
except Exception:  # noqa: BLE001
    _xr = None


def _is_xr(x) -> bool:
    return _xr is not None and isinstance(x, _xr.DataArray)


class DataArray:
    """values + dim names (+ optional 1-D coords per dim and scalar coords)."""

    __array_priority__ = 100

    def __init__(self, data, dims: Sequence[Hashable] | Hashable | None = None, coords: Mapping | None = None,
                 name: str | None = None, attrs: Mapping | None = None):
        if isinstance(data, DataArray):
            dims = data.dims if dims is None else dims
            coords = data.coords if coords is None else coords
            name = data.name if name is None else name
            data = data.values
        values = np.asarray(data)
        if dims is None:
            dims = tuple(f"dim_{i}" for i in range(values.ndim))
        elif isinstance(dims, (str, bytes)) or not isinstance(dims, Sequence):
            dims = (dims,)
        dims = tuple(dims)
        if len(dims) != values.ndim:
            raise ValueError(f"dims {dims} do not match array of shape {values.shape}")
        if len(set(dims)) != len(dims):
            raise ValueError(f"duplicate dims {dims}")
        self.values = values
        self.dims = dims
        self.name = name
        self.attrs = dict(attrs or {})
        # coords: name -> (dims, ndarray); scalars have dims ()
        self._coords: dict[Hashable, tuple[tuple, np.ndarray]] = {}
        for k, v in (coords or {}).items():
            self._set_coord(k, v)

    def _set_coord(self, k, v):
        if isinstance(v, tuple) and len(v) == 2 and not np.isscalar(v[0]) and isinstance(v[0], (tuple, list, str)):
            cd, cv = v
            cd = (cd,) if isinstance(cd, str) else tuple(cd)
            cv = np.asarray(cv.values if isinstance(cv, DataArray) else cv)
        elif isinstance(v, DataArray):
            cd, cv = v.dims, v.values
        else:
            cv = np.asarray(v)
            if cv.ndim == 0:
                cd = ()
            elif cv.ndim == 1 and k in self.dims:
                cd = (k,)
            elif cv.ndim == 1:
                match = [d for d in self.dims if self.sizes[d] == cv.shape[0]]
                if not match:
                    return
                cd = (match[0],)
            else:
                return
        if any(d not in self.dims or self.sizes[d] != n for d, n in zip(cd, cv.shape)):
            return
        self._coords[k] = (cd, cv)

    @property
    def coords(self) -> dict:
        """name -> values (ndarray); index coords are those whose name is a dim."""
        return {k: v for k, (_, v) in self._coords.items()}

    # ---- basic protocol ---------------------------------------------------
    @property
    def shape(self):
        return self.values.shape

    @property
    def ndim(self):
        return self.values.ndim

    @property
    def dtype(self):
        return self.values.dtype

    @property
    def size(self):
        return self.values.size

    @property
    def sizes(self) -> dict:
        return dict(zip(self.dims, self.values.shape))

    def __len__(self):
        return len(self.values)

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self.values, dtype=dtype)

    def __float__(self):
        return float(self.values)

    def item(self):
        return self.values.item()

    def __repr__(self):
        dims = ", ".join(f"{d}: {n}" for d, n in self.sizes.items())
        return f"<DataArray ({dims})>\n{self.values!r}"

    def copy(self):
        out = DataArray(self.values.copy(), self.dims, None, self.name, self.attrs)
        out._coords = dict(self._coords)
        return out

    def to_xarray(self):
        if _xr is None:
            raise ImportError("xarray is not installed")
        coords = {k: (cd, cv) if cd else cv for k, (cd, cv) in self._coords.items()}
        return _xr.DataArray(self.values, dims=self.dims, coords=coords, name=self.name)

    def _new(self, values, dims, drop=()):
        out = DataArray(values, dims, None, self.name, self.attrs)
        out._inherit(self._coords, drop)
        if hasattr(self, "_levels"):
            out._levels = self._levels
        return out

    def _inherit(self, coords, drop=()):
        for k, (cd, cv) in coords.items():
            if k in drop or k in self._coords:
                continue
            if all(d in self.dims and self.sizes[d] == n for d, n in zip(cd, cv.shape)):
                self._coords[k] = (cd, cv)

    def __getitem__(self, key):
        if isinstance(key, (str, bytes)):
            if key in self._coords:
                cd, cv = self._coords[key]
                return DataArray(cv, cd)
            if key in self.dims:
                return DataArray(np.arange(self.sizes[key]), (key,))
            raise KeyError(key)
        if not isinstance(key, tuple):
            key = (key,)
        idx = dict(zip(self.dims, key))
        return self.isel(idx)

    # ---- indexing ---------------------------------------------------------
    def isel(self, indexers: Mapping | None = None, drop: bool = False, **kw):
        ind = dict(indexers or {})
        ind.update(kw)
        for d in ind:
            if d not in self.dims:
                raise ValueError(f"dimension {d!r} not in {self.dims}")
        # labelled fancy indexers (DataArray of indices) -> the indexed dim is replaced by theirs
        out_vals = self.values
        out_dims = list(self.dims)
        new_coords = dict(self._coords)

        def index_coords(d, sel, scalar):
            for k in list(new_coords):
                cd, cv = new_coords[k]
                if d in cd:
                    ax_c = cd.index(d)
                    if scalar:
                        if drop and k == d:
                            new_coords.pop(k)
                        else:
                            new_coords[k] = (cd[:ax_c] + cd[ax_c + 1:], np.take(cv, int(sel), axis=ax_c))
                    elif isinstance(sel, slice):
                        new_coords[k] = (cd, cv[(slice(None),) * ax_c + (sel,)])
                    else:
                        new_coords[k] = (cd, np.take(cv, sel, axis=ax_c))

        # process from last axis to first so axis numbers stay valid
        for d in sorted(ind, key=lambda k: -self.dims.index(k)):
            ax = out_dims.index(d)
            sel = ind[d]
            if isinstance(sel, DataArray) or _is_xr(sel):
                sel = as_labelled(sel)
                out_vals = np.take(out_vals, sel.values, axis=ax)
                out_dims = out_dims[:ax] + list(sel.dims) + out_dims[ax + 1:]
                for k in [k for k, (cd, _) in new_coords.items() if d in cd]:
                    new_coords.pop(k)
            elif isinstance(sel, slice):
                out_vals = out_vals[(slice(None),) * ax + (sel,)]
                index_coords(d, sel, False)
            elif np.ndim(sel) == 0:
                out_vals = np.take(out_vals, int(sel), axis=ax)
                out_dims.pop(ax)
                index_coords(d, sel, True)
            else:
                sel = np.asarray(sel)
                out_vals = np.take(out_vals, sel, axis=ax)
                index_coords(d, sel, False)
        res = DataArray(out_vals, tuple(out_dims), None, self.name, self.attrs)
        res._inherit(new_coords)
        if hasattr(self, "_levels"):
            res._levels = self._levels
        return res

    def sel(self, indexers: Mapping | None = None, drop: bool = False, **kw):
        ind = dict(indexers or {})
        ind.update(kw)
        pos = {}
        for d, lab in ind.items():
            if d not in self.dims:
                raise ValueError(f"dimension {d!r} not in {self.dims}")
            c = self._coords[d][1] if d in self._coords and self._coords[d][0] == (d,) else None
            if c is None or np.ndim(c) != 1:
                pos[d] = lab  # no labels: positions
                continue
            if isinstance(lab, slice):
                raise NotImplementedError("label slices")
            labs = np.atleast_1d(np.asarray(lab.values if isinstance(lab, DataArray) else lab))
            where = []
            for v in labs:
                hit = np.nonzero(c == v)[0]
                if hit.size == 0:
                    raise KeyError(v)
                where.append(int(hit[0]))
            pos[d] = where[0] if np.ndim(lab) == 0 else np.asarray(where)
        return self.isel(pos, drop=drop)

    def transpose(self, *dims, missing_dims="raise"):
        if not dims:
            dims = self.dims[::-1]
        if Ellipsis in dims:
            i = dims.index(Ellipsis)
            given = [d for d in dims if d is not Ellipsis]
            rest = [d for d in self.dims if d not in given]
            dims = tuple(dims[:i]) + tuple(rest) + tuple(dims[i + 1:])
        dims = tuple(d for d in dims if d in self.dims) if missing_dims == "ignore" else tuple(dims)
        if set(dims) != set(self.dims) or len(dims) != len(self.dims):
            raise ValueError(f"{dims} must be a permutation of {self.dims}")
        return self._new(np.transpose(self.values, [self.dims.index(d) for d in dims]), dims)

    def rename(self, names: Mapping | None = None, **kw):
        m = dict(names or {})
        m.update(kw)
        dims = tuple(m.get(d, d) for d in self.dims)
        out = DataArray(self.values, dims, None, self.name, self.attrs)
        out._coords = {m.get(k, k): (tuple(m.get(d, d) for d in cd), cv) for k, (cd, cv) in self._coords.items()}
        return out

    def expand_dims(self, dim, axis=0):
        vals = np.expand_dims(self.values, axis)
        dims = list(self.dims)
        dims.insert(axis, dim)
        return self._new(vals, tuple(dims))

    def assign_coords(self, coords: Mapping | None = None, **kw):
        out = self.copy()
        c = dict(coords or {})
        c.update(kw)
        for k, v in c.items():
            out._coords.pop(k, None)
            out._set_coord(k, v)
        return out

    def drop_vars(self, names, errors="raise"):
        names = [names] if isinstance(names, (str, bytes)) else list(names)
        out = self.copy()
        for n in names:
            out._coords.pop(n, None)
        return out

    def astype(self, dtype):
        return self._new(self.values.astype(dtype), self.dims)

    # ---- stacking (the slice of MultiIndex behaviour stack.py needs) -------
    def stack(self, dimensions: Mapping | None = None, **kw):
        """Flatten groups of dims into new trailing dims; every flattened dim
        leaves a level coordinate of its name on the new dim (positions when
        it had no labels)."""
        groups = dict(dimensions or {})
        groups.update(kw)
        out = self
        for new, ds in groups.items():
            ds = (ds,) if isinstance(ds, (str, bytes)) else tuple(ds)
            if new in out.dims:
                raise ValueError(f"{new!r} conflicts with existing {out.dims}")
            keep = tuple(d for d in out.dims if d not in ds)
            t = out.transpose(*keep, *ds)
            sizes = [t.sizes[d] for d in ds]
            vals = t.values.reshape(*[t.sizes[d] for d in keep], int(np.prod(sizes, dtype=np.int64)))
            res = DataArray(vals, (*keep, new), None, out.name, out.attrs)
            res._inherit({k: v for k, v in t._coords.items() if not set(v[0]) & set(ds)})
            grids = np.meshgrid(*[t._coords[d][1] if d in t._coords and t._coords[d][0] == (d,) else np.arange(n)
                                  for d, n in zip(ds, sizes)], indexing="ij")
            for d, g in zip(ds, grids):
                res._coords[d] = ((new,), g.reshape(-1))
            res._levels = {**getattr(out, "_levels", {}), new: ds}
            out = res
        return out

    @property
    def indexes(self) -> dict:
        """dim -> index; a stacked dim gives a ``LevelIndex`` of tuples."""
        out = {}
        levels = getattr(self, "_levels", {})
        for d in self.dims:
            if d in levels and all(n in self._coords for n in levels[d]):
                out[d] = LevelIndex(levels[d], [self._coords[n][1] for n in levels[d]])
            elif d in self._coords and self._coords[d][0] == (d,):
                out[d] = self._coords[d][1]
        return out

    def groupby(self, dim):
        """Iterate ``(position, slice)`` along ``dim`` (one group per element)."""
        for i in range(self.sizes[dim]):
            yield i, self.isel({dim: i})

    def pipe(self, func, *args, **kwargs):
        return func(self, *args, **kwargs)

    # ---- reductions -------------------------------------------------------
    def _reduce(self, fn, dim=None, **kw):
        if dim is None:
            axes = None
            dims = ()
        else:
            ds = [dim] if isinstance(dim, (str, bytes)) or not isinstance(dim, Sequence) else list(dim)
            for d in ds:
                if d not in self.dims:
                    raise ValueError(f"dimension {d!r} not in {self.dims}")
            axes = tuple(self.dims.index(d) for d in ds)
            dims = tuple(d for d in self.dims if d not in ds)
        return self._new(fn(self.values, axis=axes, **kw), dims)

    def sum(self, dim=None):
        return self._reduce(np.sum, dim)

    def mean(self, dim=None):
        return self._reduce(np.mean, dim)

    def std(self, dim=None, ddof=0):
        return self._reduce(np.std, dim, ddof=ddof)

    def var(self, dim=None, ddof=0):
        return self._reduce(np.var, dim, ddof=ddof)

    def max(self, dim=None):
        return self._reduce(np.max, dim)

    def min(self, dim=None):
        return self._reduce(np.min, dim)

    def cumsum(self, dim):
        ax = self.dims.index(dim)
        return self._new(np.cumsum(self.values, axis=ax), self.dims)

    # ---- arithmetic with broadcasting by name -------------------------------
    def _binary(self, other, op, reflexive=False):
        if _is_xr(other):
            other = as_labelled(other)
        if isinstance(other, DataArray):
            dims = list(self.dims) + [d for d in other.dims if d not in self.dims]
            a = _expand(self, dims)
            b = _expand(other, dims)
            coords = {**other._coords, **self._coords}
        else:
            dims = list(self.dims)
            a, b = self.values, other
            coords = self._coords
        vals = op(b, a) if reflexive else op(a, b)
        out = DataArray(vals, tuple(dims), None, self.name)
        out._inherit(coords)
        return out

    def __add__(self, o): return self._binary(o, np.add)
    def __radd__(self, o): return self._binary(o, np.add, True)
    def __sub__(self, o): return self._binary(o, np.subtract)
    def __rsub__(self, o): return self._binary(o, np.subtract, True)
    def __mul__(self, o): return self._binary(o, np.multiply)
    def __rmul__(self, o): return self._binary(o, np.multiply, True)
    def __truediv__(self, o): return self._binary(o, np.true_divide)
    def __rtruediv__(self, o): return self._binary(o, np.true_divide, True)
    def __pow__(self, o): return self._binary(o, np.power)
    def __rpow__(self, o): return self._binary(o, np.power, True)
    def __neg__(self): return self._new(-self.values, self.dims)
    def __abs__(self): return self._new(np.abs(self.values), self.dims)

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        if method != "__call__" or kwargs.get("out") is not None:
            return NotImplemented
        if len(inputs) == 1:
            return self._new(ufunc(self.values, **kwargs), self.dims)
        if len(inputs) == 2:
            a, b = inputs
            if a is self:
                return self._binary(b, lambda x, y: ufunc(x, y, **kwargs))
            return self._binary(a, lambda x, y: ufunc(x, y, **kwargs), True)
        return NotImplemented


class LevelIndex:
    """Names + per-level label arrays of a stacked dim; ``values`` are tuples."""

    def __init__(self, names, arrays):
        self.names = tuple(names)
        self.arrays = [np.asarray(a) for a in arrays]

    @property
    def values(self):
        return list(zip(*[a.tolist() for a in self.arrays]))

    def __len__(self):
        return len(self.arrays[0])

    def __iter__(self):
        return iter(self.values)


def _expand(a: DataArray, dims) -> np.ndarray:
    """values of `a` transposed/expanded to broadcast against `dims`."""
    order = [d for d in dims if d in a.dims]
    v = np.transpose(a.values, [a.dims.index(d) for d in order])
    shape = [a.sizes[d] if d in a.dims else 1 for d in dims]
    return v.reshape(shape)


class Dataset:
    """A named collection of labelled arrays sharing dims (the part of xarray.Dataset the reference's data classes
    accept for ``xv``, reference data.py:347-350): mapping access, ``data_vars``, ``map``."""

    def __init__(self, data_vars: Mapping[Hashable, Any] | None = None, attrs: Mapping | None = None):
        self._vars: dict[Hashable, DataArray] = {}
        for k, v in (data_vars or {}).items():
            self[k] = v
        self.attrs = dict(attrs or {})

    def __setitem__(self, key, value):
        if is_labelled(value):
            v = as_labelled(value)
            self._vars[key] = v if v.name == key else DataArray(v.values, v.dims, v.coords, key, v.attrs)
        elif hasattr(value, "dims") and hasattr(value, "sizes"):  # e.g. moments.DeviceDataArray (samples resident in HBM)
            self._vars[key] = value
        else:
            raise TypeError(f"Dataset variable {key!r} must be a labelled array")

    def __getitem__(self, key) -> DataArray:
        return self._vars[key]

    def __iter__(self):
        return iter(self._vars)

    def __len__(self):
        return len(self._vars)

    def __contains__(self, key):
        return key in self._vars

    def keys(self):
        return self._vars.keys()

    def items(self):
        return self._vars.items()

    def values(self):
        return self._vars.values()

    @property
    def data_vars(self) -> dict:
        return dict(self._vars)

    @property
    def sizes(self) -> dict:
        out: dict = {}
        for v in self._vars.values():
            for d, n in v.sizes.items():
                if out.setdefault(d, n) != n:
                    raise ValueError(f"conflicting sizes for dimension {d!r}")
        return out

    @property
    def dims(self) -> tuple:
        return tuple(self.sizes)

    def map(self, func, *args, **kwargs) -> "Dataset":
        return Dataset({k: func(v, *args, **kwargs) for k, v in self._vars.items()}, self.attrs)

    def __repr__(self):
        body = "\n".join(f"    {k}: {v.dims} {v.shape}" for k, v in self._vars.items())
        return f"<thermoextrap_amd.xrlite.Dataset>\n{body}"


def is_dataset(x) -> bool:
    return isinstance(x, Dataset) or (_xr is not None and isinstance(x, _xr.Dataset))


def as_dataset(x) -> Dataset:
    if isinstance(x, Dataset):
        return x
    return Dataset({k: as_labelled(v) for k, v in x.data_vars.items()}, getattr(x, "attrs", None))


def as_labelled(x, dims=None, name=None) -> DataArray:
    """DataArray from ours / xarray's / array-like (dims required for the latter)."""
    if isinstance(x, DataArray):
        return x
    if _is_xr(x):
        coords = {k: (tuple(v.dims), np.asarray(v.values)) for k, v in x.coords.items()}
        return DataArray(np.asarray(x.values), tuple(x.dims), coords, x.name)
    return DataArray(np.asarray(x), dims, name=name)


def is_labelled(x) -> bool:
    return isinstance(x, DataArray) or _is_xr(x)


def concat(objs, dim, coords=None):
    """Stack labelled arrays along a NEW dim (name, or a 1-D DataArray / (name, values) giving its coordinate)."""
    objs = [as_labelled(o) for o in objs]
    cvals = None
    if isinstance(dim, DataArray):
        cvals = dim.values
        dim = dim.dims[0] if dim.dims else dim.name
    elif hasattr(dim, "name") and hasattr(dim, "values"):  # pandas Index
        cvals = np.asarray(dim.values)
        dim = dim.name
    first = objs[0]
    if dim in first.dims:
        ax = first.dims.index(dim)
        vals = np.concatenate([o.transpose(*first.dims).values for o in objs], axis=ax)
        return DataArray(vals, first.dims, None, first.name)
    alld = list(first.dims)
    for o in objs[1:]:
        alld += [d for d in o.dims if d not in alld]
    vals = np.stack([np.broadcast_to(_expand(o, alld), [max(p.sizes.get(d, 1) for p in objs) for d in alld]) for o in objs], axis=0)
    out = DataArray(vals, (dim, *alld), None, first.name)
    out._inherit(first._coords)
    if cvals is not None:
        out._set_coord(dim, np.asarray(cvals))
    return out


def assert_allclose(a, b, rtol=1e-7, atol=0.0):
    """xr.testing.assert_allclose for labelled arrays (dims must match as sets; b is aligned to a)."""
    a, b = as_labelled(a), as_labelled(b)
    if set(a.dims) != set(b.dims):
        raise AssertionError(f"dims differ: {a.dims} vs {b.dims}")
    np.testing.assert_allclose(a.values, b.transpose(*a.dims).values, rtol=rtol, atol=atol)
