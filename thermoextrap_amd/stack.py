"""Mean/variance stacking of replicate derivatives for GP regression input
(reference stack.py:15-216, 519-660): the consumer of ``StateCollection.resample``
+ ``ExtrapModel.derivs`` that turns ``(alpha, order, rep, ...)`` into the
``X = (alpha, order)``, ``Y = (mean, var)`` arrays GPflow is trained on.

The replicate derivatives arrive from txm_eval_poly; the reductions here run
over ``nrep`` numbers per output element on the host (kilobytes).  GPflow itself
stays external, as in gpr_input.py.
"""

from __future__ import annotations

import numpy as np

from .models import StateCollection
from .xrlite import DataArray, as_labelled, concat

__all__ = ["GPRData", "apply_reduction", "multiindex_to_array", "stack_dataarray", "states_derivs_concat",
           "to_mean_var", "wrap_like_dataarray"]


def stack_dataarray(da, x_dims, y_dims=None, xstack_dim="xstack", ystack_dim="ystack", stats_dim=None,
                    policy="infer"):
    """Flatten ``x_dims`` -> ``xstack_dim`` and the remaining (or given) ``y_dims``
    -> ``ystack_dim``; ``stats_dim`` is moved last (reference stack.py:15-84)."""
    da = as_labelled(da)
    for name in (xstack_dim, ystack_dim):
        if name in da.dims:
            raise ValueError(f"{name} conflicts with existing {da.dims}")
    x_dims = (x_dims,) if isinstance(x_dims, str) else tuple(x_dims)
    if isinstance(y_dims, str):
        y_dims = (y_dims,)
    elif y_dims is None:
        skip = set(x_dims) | ({stats_dim} if isinstance(stats_dim, str) else set(stats_dim or ()))
        y_dims = tuple(d for d in da.dims if d not in skip)
    groups = {xstack_dim: x_dims}
    if len(y_dims) > 0:
        groups[ystack_dim] = tuple(y_dims)
    if policy == "raise":
        for d in x_dims:
            if d not in da.coords:
                raise ValueError(f"da.coords[{d}] not set")
    out = da.stack(groups)
    if stats_dim is not None:
        tail = (stats_dim,) if isinstance(stats_dim, str) else tuple(stats_dim)
        out = out.transpose(..., *tail)
    return out


def wrap_like_dataarray(x, da):
    """Array ``x`` with the labels of ``da``."""
    da = as_labelled(da)
    out = DataArray(np.asarray(x), da.dims, None, da.name, da.attrs)
    out._inherit(da._coords)
    return out


def multiindex_to_array(idx):
    """Stacked index -> ``[n, n_levels]`` array (reference stack.py:99-101)."""
    return np.array(list(idx.values))


def apply_reduction(da, dim, funcs, concat=True, concat_dim=None, concat_kws=None, **kws):  # noqa: ARG001
    """Several reductions of one array (names of methods or callables) (reference stack.py:104-154)."""
    da = as_labelled(da)
    if not isinstance(funcs, (tuple, list)):
        funcs = [funcs]
    out = [f(da, dim=dim, **kws) if callable(f) else getattr(da, f)(dim=dim, **kws) for f in funcs]
    if len(out) == 1:
        return out[0]
    if concat_dim is not None:
        return _concat(out, concat_dim, **(concat_kws or {}))
    return out


def _concat(objs, dim, **kws):
    return concat(list(objs), dim=dim, **kws)


def to_mean_var(da, dim, concat_dim=None, concat_kws=None, **kws):
    """``concat([mean(dim), var(dim)])`` along ``stats`` = ["mean", "var"] (reference stack.py:157-183)."""
    da = as_labelled(da)
    if concat_dim is None:
        concat_dim = DataArray(np.array(["mean", "var"]), "stats")
    return _concat((da.mean(dim, **kws), da.var(dim, **kws)), concat_dim, **(concat_kws or {}))


def states_derivs_concat(states, dim=None, concat_kws=None, **kws):
    """``concat([s.derivs(norm=False) for s in states])`` along ``alpha_name`` (reference stack.py:186-216)."""
    if dim is None:
        dim = DataArray(np.asarray(states.alpha0, dtype=float), states.alpha_name)
    kws.setdefault("norm", False)
    return _concat((s.derivs(**kws) for s in states), dim, **(concat_kws or {}))


class _Stacked:
    """What ``GPRData`` and ``StackedDerivatives`` share: X/Y arrays out of the stacked view."""

    def stacked(self, order=None):
        if order is None:
            order = self.order
        if ("stacked", order) not in self._cache:
            self._cache[("stacked", order)] = self._stacked(order)
        return self._cache[("stacked", order)]

    def array_data(self, order=None):
        """``X[n_x, len(x_dims)]`` and a list over the stacked y elements of ``Y[n_x, 2]``."""
        st = self.stacked(order=order)
        xdata = multiindex_to_array(st.indexes[self.xstack_dim])
        if self.ystack_dim in st.dims:
            ydata = [g.values for _, g in st.groupby(self.ystack_dim)]
        else:
            ydata = [st.values]
        return xdata, ydata

    def xindexer_from_arrays(self, **kwargs):
        """Rows ``(*x_dims[:-1], order=0)`` to look predictions up by name."""
        names = list(self.x_dims[:-1])
        if set(kwargs) != set(names):
            raise ValueError(f"need exactly {names}")
        cols = [np.atleast_1d(np.asarray(kwargs[n])) for n in names]
        return [(*vals, 0) for vals in zip(*[c.tolist() for c in cols])]

    def xindexer_from_dataframe(self, df):
        if set(df.columns) != set(self.x_dims[:-1]):
            raise ValueError
        return self.xindexer_from_arrays(**{c: df[c].to_numpy() for c in df.columns})


class StackedDerivatives(_Stacked):
    """Mean/variance of derivatives already computed (reference stack.py:219-516)."""

    def __init__(self, da, x_dims, y_dims=None, xstack_dim="xstack", ystack_dim="ystack", stats_dim="stats",
                 policy="infer"):
        self.da = as_labelled(da)
        self.x_dims = [x_dims] if isinstance(x_dims, str) else list(x_dims)
        self.y_dims = [y_dims] if isinstance(y_dims, str) else y_dims
        self.xstack_dim, self.ystack_dim, self.stats_dim, self.policy = xstack_dim, ystack_dim, stats_dim, policy
        self._cache: dict = {}

    @property
    def order_dim(self):
        return self.x_dims[-1]

    @property
    def order(self):
        return self.da.sizes[self.order_dim] - 1

    @property
    def alpha_name(self):
        return self.x_dims[0]

    def _stacked(self, order):
        da = self.da.isel({self.order_dim: slice(None, order + 1)})
        return stack_dataarray(da, x_dims=self.x_dims, y_dims=self.y_dims, xstack_dim=self.xstack_dim,
                               ystack_dim=self.ystack_dim, stats_dim=self.stats_dim, policy=self.policy)

    @classmethod
    def from_mean_var(cls, mean, var, x_dims, y_dims=None, xstack_dim="xstack", ystack_dim="ystack",
                      stats_dim="stats", policy="infer", concat_kws=None):
        da = _concat((mean, var), DataArray(np.array(["mean", "var"]), stats_dim), **(concat_kws or {}))
        return cls(da, x_dims, y_dims, xstack_dim, ystack_dim, stats_dim, policy)

    @classmethod
    def from_derivs(cls, derivs, x_dims, reduce_dim="rep", y_dims=None, xstack_dim="xstack", ystack_dim="ystack",
                    stats_dim="stats", policy="infer", concat_kws=None):
        da = to_mean_var(derivs, reduce_dim, DataArray(np.array(["mean", "var"]), stats_dim), concat_kws)
        return cls(da, x_dims, y_dims, xstack_dim, ystack_dim, stats_dim, policy)

    @classmethod
    def from_states(cls, states, x_dims=None, resample=False, resample_kws=None, map_func="derivs", map_kws=None,
                    reduce_dim="rep", concat_dim=None, concat_kws=None, **kws):
        if resample:
            states = states.resample(**(resample_kws or {}))
        if x_dims is None:
            x_dims = [states.alpha_name, "order"]
        derivs = states.map_concat(map_func, concat_dim, concat_kws, **{"norm": False, **(map_kws or {})})
        return cls.from_derivs(derivs, x_dims=x_dims, reduce_dim=reduce_dim, **kws)


class GPRData(StateCollection, _Stacked):
    """State collection whose replicate derivatives are served as GP training arrays (reference stack.py:519-660)."""

    def __init__(self, states, x_dims=None, y_dims=None, xstack_dim="xstack", ystack_dim="ystack",
                 stats_dim="stats", reduce_dim="rep", deriv_kws=None, kws=None):
        if x_dims is None:
            x_dims = [states[0].alpha_name, "order"]
        self.x_dims = list(x_dims)
        self.y_dims = y_dims
        self.xstack_dim, self.ystack_dim, self.stats_dim = xstack_dim, ystack_dim, stats_dim
        self.reduce_dim = reduce_dim
        self.deriv_kws = dict(deriv_kws or {})
        super().__init__(states, kws=kws)

    def resample(self, sampler, **kws):
        base = StateCollection(self.states).resample(sampler, **kws)
        return type(self)(base.states, x_dims=self.x_dims, y_dims=self.y_dims, xstack_dim=self.xstack_dim,
                          ystack_dim=self.ystack_dim, stats_dim=self.stats_dim, reduce_dim=self.reduce_dim,
                          deriv_kws=self.deriv_kws)

    @property
    def order_dim(self):
        return self.x_dims[-1]

    def _stacked(self, order):
        kws = dict(self.deriv_kws, order_dim=self.order_dim)
        d = states_derivs_concat(self, order=order, **kws)
        mv = to_mean_var(d, dim=self.reduce_dim, concat_dim=DataArray(np.array(["mean", "var"]), self.stats_dim))
        return stack_dataarray(mv, x_dims=self.x_dims, y_dims=self.y_dims, xstack_dim=self.xstack_dim,
                               ystack_dim=self.ystack_dim, stats_dim=self.stats_dim, policy="infer")
