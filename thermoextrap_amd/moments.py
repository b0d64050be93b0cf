"""Mirror of the slice of the cmomy 0.24 API that thermoextrap calls
(/root/reference/src/thermoextrap/data.py, lnpi.py, volume.py, models.py:626-630;
inventory in SURVEY.md 8(b)), executed by libtxmom on the MI355X.

Same names and argument meaning as cmomy so that the data classes read like the
reference's:  wrap_reduce_vals, wrap_resample_vals, factory_sampler,
IndexSampler, CentralMomentsData, select_moment, assign_moment, wrap,
convert.{moments_type, moments_to_comoments, comoments_to_moments},
random.{default_rng, validate_rng}.

Moment states live on the device (torch CUDA tensors); labelled host views are
made on demand.  There is no CPU implementation behind these calls.
"""

from __future__ import annotations

from collections.abc import Mapping, Sequence
from types import SimpleNamespace

import numpy as np
import torch

from . import engine
from .xrlite import DataArray, as_labelled, is_labelled

MISSING = None

# element count (nrep * ndat) up to which {"nrep": n} draws explicit numpy
# indices exactly as the reference does; above it the device multinomial
# sampler is used (the reference would need 16 bytes per element of tables).
EXPLICIT_SAMPLER_MAX = 1 << 27


# ---------------------------------------------------------------------------
# random (cmomy.random)
# ---------------------------------------------------------------------------
_GLOBAL_RNG: np.random.Generator | None = None


def default_rng(seed=None) -> np.random.Generator:
    """Set and return the module-global Generator (cmomy.random.default_rng;
    behaviour verified against the reference's seeded notebooks)."""
    global _GLOBAL_RNG
    if seed is not None or _GLOBAL_RNG is None:
        _GLOBAL_RNG = np.random.default_rng(seed)
    return _GLOBAL_RNG


def validate_rng(rng=None) -> np.random.Generator:
    if rng is None:
        return default_rng()
    if isinstance(rng, np.random.Generator):
        return rng
    return np.random.default_rng(rng)


random = SimpleNamespace(default_rng=default_rng, validate_rng=validate_rng)


# ---------------------------------------------------------------------------
# device-resident labelled values
# ---------------------------------------------------------------------------
class DeviceDataArray:
    """A torch CUDA tensor with dim names: how sample arrays that already live in
    HBM (or are too big to bounce through numpy) enter the labelled API."""

    def __init__(self, tensor: torch.Tensor, dims: Sequence):
        if tensor.dim() != len(tuple(dims)):
            raise ValueError("dims do not match tensor rank")
        self.tensor = tensor if tensor.is_cuda else tensor.cuda()
        if self.tensor.dtype != torch.float64:
            self.tensor = self.tensor.double()
        self.dims = tuple(dims)

    @property
    def sizes(self):
        return dict(zip(self.dims, self.tensor.shape))

    @property
    def shape(self):
        return tuple(self.tensor.shape)

    @property
    def values(self):
        return self.tensor.cpu().numpy()

    def __len__(self):
        return self.tensor.shape[0]


def _dev_and_dims(x, dims=None) -> tuple[torch.Tensor, tuple]:
    """(float64 CUDA tensor, dims) of a labelled / device / raw array."""
    if isinstance(x, DeviceDataArray):
        return x.tensor, x.dims
    if is_labelled(x):
        x = as_labelled(x)
        # device copy cached on the object, keyed by the identity of its buffer
        cache = getattr(x, "_txm_dev", None)
        if cache is None or cache[0] is not x.values:
            cache = (x.values, engine.to_device(x.values))
            x._txm_dev = cache
        return cache[1], x.dims
    if isinstance(x, torch.Tensor):
        if dims is None:
            raise TypeError("a bare tensor needs dims")
        return x.to(device="cuda", dtype=torch.float64), tuple(dims)
    raise TypeError(f"type {type(x)} must be a DataArray (or DeviceDataArray)")


def _resolve_dim(dims, dim, axis, default=None):
    if dim is not None:
        if dim not in dims:
            raise ValueError(f"dimension {dim!r} not found in array dimensions {dims}")
        return dim
    if axis is not None:
        return dims[axis]
    if default is not None:
        return default
    raise ValueError("must specify dim or axis")


# ---------------------------------------------------------------------------
# sampler (cmomy.resample.factory_sampler / IndexSampler)
# ---------------------------------------------------------------------------
class IndexSampler:
    """Resampling plan: explicit `indices`/`freq` ("parity mode", identical
    semantics and draws to the reference) or a counter-based device stream
    ("scale mode", txm_sampler_*)."""

    def __init__(self, *, indices=None, freq=None, device_sampler: engine.DeviceSampler | None = None,
                 ndat: int | None = None, rep_dim: str = "rep"):
        if indices is None and freq is None and device_sampler is None:
            raise ValueError("need indices, freq or a device sampler")
        self._indices = None if indices is None else np.asarray(getattr(indices, "values", indices), dtype=np.int64)
        self._freq_host = None if freq is None else np.asarray(getattr(freq, "values", freq), dtype=np.int64)
        self._freq_dev: torch.Tensor | None = None
        self.device_sampler = device_sampler
        self.rep_dim = rep_dim
        if self._indices is not None:
            if self._indices.ndim != 2:
                raise ValueError("indices must be (nrep, nsamp)")
            self._ndat = int(ndat) if ndat is not None else self._indices.shape[1]
            self._nrep = self._indices.shape[0]
        elif self._freq_host is not None:
            if self._freq_host.ndim != 2:
                raise ValueError("freq must be (nrep, ndat)")
            self._nrep, self._ndat = self._freq_host.shape
        else:
            self._nrep, self._ndat = device_sampler.nrep, device_sampler.ndat

    @property
    def nrep(self) -> int:
        return self._nrep

    @property
    def ndat(self) -> int:
        return self._ndat

    @property
    def is_device(self) -> bool:
        return self.device_sampler is not None

    @property
    def indices(self) -> np.ndarray:
        """(nrep, nsamp) int64.  Explicit samplers return what they were given; for
        freq/device samplers an equivalent index table is materialised (small sizes only)."""
        if self._indices is None:
            f = self.freq
            if f.size > EXPLICIT_SAMPLER_MAX:
                raise MemoryError("index table too large to materialise; use the freq/device path")
            self._indices = np.stack([np.repeat(np.arange(self._ndat), row) for row in f])
        return self._indices

    @property
    def freq(self) -> np.ndarray:
        if self._freq_host is None:
            self._freq_host = self.freq_device().cpu().numpy()
        return self._freq_host

    def freq_device(self) -> torch.Tensor:
        """(nrep, ndat) int64 on the device (built by txm_indices_to_freq / txm_sampler_freq)."""
        if self._freq_dev is None:
            if self._freq_host is not None:
                self._freq_dev = torch.as_tensor(self._freq_host).cuda()
            elif self._indices is not None:
                self._freq_dev = engine.indices_to_freq(torch.as_tensor(self._indices).cuda(), self._ndat)
            else:
                self._freq_dev = self.device_sampler.freq()
        return self._freq_dev


def factory_sampler(sampler=None, *, indices=None, freq=None, ndat=None, nrep=None, nsamp=None, rng=None,
                    data=None, dim=None, axis=None, mom_ndim=None, mom_dims=None, rep_dim="rep",
                    parallel=None, device: bool | None = None, seed: int | None = None,
                    rep0: int = 0) -> IndexSampler:
    """cmomy.factory_sampler.  `sampler` may be an IndexSampler, a mapping of the
    keyword arguments, or an array of indices.  `device=True` (or a table above
    EXPLICIT_SAMPLER_MAX elements) selects the device multinomial sampler; `rep0` (extension, device sampler
    only) makes replicate r of the result replicate rep0 + r of the stream of `seed` -- rows rep0 .. rep0 + nrep
    of a larger table, bit for bit (multi-GPU shards, txm_sampler_spec.rep0)."""
    del parallel, mom_ndim
    if isinstance(sampler, IndexSampler):
        return sampler
    if isinstance(sampler, Mapping):
        kw = dict(sampler)
        return factory_sampler(None, indices=kw.get("indices", indices), freq=kw.get("freq", freq),
                               ndat=kw.get("ndat", ndat), nrep=kw.get("nrep", nrep), nsamp=kw.get("nsamp", nsamp),
                               rng=kw.get("rng", rng), data=data, dim=dim, axis=axis, mom_dims=mom_dims,
                               rep_dim=kw.get("rep_dim", rep_dim), device=kw.get("device", device),
                               seed=kw.get("seed", seed), rep0=kw.get("rep0", rep0))
    if sampler is not None:  # array of indices
        indices = sampler
    if ndat is None and data is not None:
        dims = data.dims
        d = _resolve_dim(dims, dim, axis)
        ndat = data.sizes[d] if hasattr(data, "sizes") else data.shape[dims.index(d)]
    if indices is not None:
        return IndexSampler(indices=indices, ndat=ndat, rep_dim=rep_dim)
    if freq is not None:
        return IndexSampler(freq=freq, rep_dim=rep_dim)
    if nrep is None or ndat is None:
        raise ValueError("need nrep and ndat (or data) to build a sampler")
    nsamp = ndat if nsamp is None else int(nsamp)
    use_device = device if device is not None else (int(nrep) * int(nsamp) > EXPLICIT_SAMPLER_MAX)
    if rep0 and not use_device:
        raise ValueError("rep0 addresses the device sampler's stream: pass device=True (numpy draws have no replicate index)")
    if use_device:
        if seed is None:
            seed = int(validate_rng(rng).integers(0, 2**63 - 1))
        return IndexSampler(device_sampler=engine.DeviceSampler(seed, int(nrep), int(ndat), 0 if nsamp == ndat else nsamp,
                                                                rep0=int(rep0)),
                            rep_dim=rep_dim)
    # the reference's draw (cmomy 0.24; verified in SURVEY App. B)
    idx = validate_rng(rng).choice(int(ndat), size=(int(nrep), nsamp), replace=True)
    return IndexSampler(indices=idx, ndat=ndat, rep_dim=rep_dim)


# ---------------------------------------------------------------------------
# CentralMomentsData
# ---------------------------------------------------------------------------
class CentralMomentsData:
    """Device-resident central (co)moment states with dim names.

    ``dims`` ends with the moment dims (1 for mom_ndim=1, 2 for mom_ndim=2); the
    state layout is cmomy's: [0,0]=weight, [1,0]=<x>, [0,1]=<u>, else central."""

    def __init__(self, data, mom_ndim: int = 2, mom_dims=None, dims=None):
        if isinstance(data, CentralMomentsData):
            data, dims, mom_ndim = data._dev, data.dims, data.mom_ndim
        if isinstance(data, torch.Tensor):
            if dims is None:
                raise ValueError("dims required with a tensor")
            self._dev = data.to(device="cuda", dtype=torch.float64)
            self.dims = tuple(dims)
            self._coords = {}
        else:
            lab = as_labelled(data)
            self._dev = engine.to_device(lab.values)
            self.dims = lab.dims
            self._coords = dict(lab._coords)
        if mom_ndim not in (1, 2):
            raise ValueError("mom_ndim must be 1 or 2")
        self.mom_ndim = mom_ndim
        if mom_dims is not None:
            mom_dims = (mom_dims,) if isinstance(mom_dims, str) else tuple(mom_dims)
            if len(mom_dims) != mom_ndim:
                raise ValueError("mom_dims does not match mom_ndim")
            if any(d not in self.dims for d in mom_dims):
                raise ValueError(f"{mom_dims} not in {self.dims}")
            order = [d for d in self.dims if d not in mom_dims] + list(mom_dims)
            if tuple(order) != self.dims:
                self._dev = self._dev.permute([self.dims.index(d) for d in order]).contiguous()
                self.dims = tuple(order)
        if mom_ndim == 2 and self._dev.shape[-2] != 2:
            raise ValueError("comoment states must be (..., 2, order+1)")
        self._host: DataArray | None = None

    # ---- views ------------------------------------------------------------
    @property
    def mom_dims(self) -> tuple:
        return self.dims[-self.mom_ndim:]

    @property
    def val_dims(self) -> tuple:
        return self.dims[: -self.mom_ndim]

    @property
    def sizes(self) -> dict:
        return dict(zip(self.dims, self._dev.shape))

    @property
    def shape(self):
        return tuple(self._dev.shape)

    @property
    def mom(self) -> tuple:
        return tuple(n - 1 for n in self._dev.shape[-self.mom_ndim:])

    @property
    def device_values(self) -> torch.Tensor:
        return self._dev

    def _label(self, t: torch.Tensor, dims=None) -> DataArray:
        out = DataArray(t.cpu().numpy(), self.dims if dims is None else dims)
        out._inherit(self._coords)
        return out

    @property
    def obj(self) -> DataArray:
        if self._host is None:
            self._host = self._label(self._dev)
        return self._host

    @property
    def values(self) -> np.ndarray:
        return self.obj.values

    def to_numpy(self):
        return self.obj.values

    def _like(self, dev, dims=None, mom_ndim=None) -> "CentralMomentsData":
        out = CentralMomentsData(dev, mom_ndim=self.mom_ndim if mom_ndim is None else mom_ndim,
                                 dims=self.dims if dims is None else dims)
        out._coords = {k: v for k, v in self._coords.items() if all(d in out.dims for d in v[0])}
        return out

    def transpose(self, *dims):
        if Ellipsis in dims:
            i = dims.index(Ellipsis)
            given = [d for d in dims if d is not Ellipsis]
            rest = [d for d in self.dims if d not in given]
            dims = tuple(dims[:i]) + tuple(rest) + tuple(dims[i + 1:])
        dims = tuple(dims)
        if tuple(dims[-self.mom_ndim:]) != self.mom_dims:
            raise ValueError("moment dims must stay last")
        return self._like(self._dev.permute([self.dims.index(d) for d in dims]).contiguous(), dims)

    # ---- raw/central views (data.py:844-852) --------------------------------
    def cmom_device(self) -> torch.Tensor:
        c = self._dev.clone()
        if self.mom_ndim == 2:
            c[..., 0, 0] = 1.0
            c[..., 1, 0] = 0.0
            if c.shape[-1] > 1:
                c[..., 0, 1] = 0.0
        else:
            c[..., 0] = 1.0
            if c.shape[-1] > 1:
                c[..., 1] = 0.0
        return c

    def rmom_device(self) -> torch.Tensor:
        r = engine.convert_cov(self._dev, False) if self.mom_ndim == 2 else engine.convert_1d(self._dev, False)
        if self.mom_ndim == 2:
            r[..., 0, 0] = 1.0
        else:
            r[..., 0] = 1.0
        return r

    def cmom(self) -> DataArray:
        """Central moments with [0,0] -> 1 and the first-order entries -> 0."""
        return self._label(self.cmom_device())

    def rmom(self) -> DataArray:
        """Raw moments <x^a u^b>, [0,0] -> 1."""
        return self._label(self.rmom_device())

    def select_moment(self, name: str) -> DataArray:
        return select_moment(self.obj, name, mom_ndim=self.mom_ndim, mom_dims=self.mom_dims)

    def weight(self) -> DataArray:
        return self.select_moment("weight")

    # ---- reductions -------------------------------------------------------
    def _canon(self, dim):
        """permute to (dim, flat other value dims, moments); returns tensor, other dims, their shape"""
        if self.mom_ndim != 2:
            raise NotImplementedError("merging of 1-D moment states")
        if dim not in self.val_dims:
            raise ValueError(f"dimension {dim!r} not found in array dimensions {self.dims}")
        others = [d for d in self.val_dims if d != dim]
        perm = [self.dims.index(dim)] + [self.dims.index(d) for d in others] + [len(self.dims) - 2, len(self.dims) - 1]
        t = self._dev.permute(perm).contiguous()
        oshape = t.shape[1:-2]
        return t.reshape(t.shape[0], -1, 2, t.shape[-1]), others, tuple(oshape)

    def reduce(self, dim=MISSING, axis=MISSING, **kw) -> "CentralMomentsData":
        """Merge the states along `dim` (CentralMomentsData.reduce, data.py:996)."""
        d = _resolve_dim(self.dims, dim, axis)
        t, others, oshape = self._canon(d)
        out = engine.resample_data(t, None, t.shape[-1] - 1)[0]
        return self._like(out.reshape(*oshape, 2, t.shape[-1]), tuple(others) + self.mom_dims)

    def push_vals(self, x, *y, weight=None, axis=MISSING, dim=MISSING, **kw) -> "CentralMomentsData":
        """cmomy ``CentralMomentsData.push_vals(x, u, weight=, dim=)``: accumulate new samples into this object IN PLACE
        (north_star's streaming accumulation; thermoextrap itself never calls it -- SURVEY 0.7).  ``x``: samples along
        ``dim`` with this object's value dims as the other dims, ``y[0]``: u along ``dim``.  A state of zeros is the empty
        accumulator.  One reduction of the chunk + one merge kernel (txm_push_vals)."""
        del kw
        if self.mom_ndim != 2 or len(y) != 1:
            raise NotImplementedError("push_vals: comoment states with exactly one y array")
        xt, xdims = _dev_and_dims(x)
        red = _resolve_dim(xdims, dim, axis)
        ut, udims = _dev_and_dims(y[0])
        if udims != (red,):
            raise NotImplementedError("push_vals: uv must be 1-D along the sample dim")
        cols = tuple(d for d in xdims if d != red)
        if cols != self.val_dims:
            raise ValueError(f"value dims of x {cols} do not match the state's {self.val_dims}")
        N = xt.shape[xdims.index(red)]
        wt = None
        if weight is not None:
            wt = _dev_and_dims(weight)[0] if (is_labelled(weight) or isinstance(weight, DeviceDataArray)) else engine.to_device(np.asarray(weight))
        x2 = xt.movedim(xdims.index(red), 0)
        x2 = x2.reshape(N, -1) if x2.dim() > 1 else x2
        st = self._dev.reshape(-1, 2, self._dev.shape[-1])
        if not st.is_contiguous() or st.data_ptr() != self._dev.data_ptr():
            raise ValueError("push_vals needs a contiguous state")
        engine.push_vals(st, x2 if x2.dim() == 2 else x2, ut.contiguous(), w=wt)
        self._host = None
        return self

    def resample_and_reduce(self, *, sampler, dim=MISSING, axis=MISSING, rep_dim="rep", parallel=None, **kw):
        """Block bootstrap (data.py:1048-1052): out dims = (other value dims with
        `dim` replaced by rep_dim in place, moments)."""
        del parallel, kw
        d = _resolve_dim(self.dims, dim, axis)
        sampler = factory_sampler(sampler, data=self, dim=d, rep_dim=rep_dim)
        if sampler.ndat != self.sizes[d]:
            raise ValueError(f"sampler.ndat={sampler.ndat} must equal size of {d!r}={self.sizes[d]}")
        t, others, oshape = self._canon(d)
        out = engine.resample_data(t, sampler.freq_device(), t.shape[-1] - 1)  # (nrep, C, 2, K)
        out = out.reshape(sampler.nrep, *oshape, 2, t.shape[-1])
        cur = (rep_dim, *others, *self.mom_dims)
        # cmomy puts rep_dim where `dim` was
        target = tuple(rep_dim if x == d else x for x in self.dims)
        return self._like(out.permute([cur.index(x) for x in target]).contiguous(), target)

    # ---- x_is_u helpers (data.py:1182-1191, 1265-1268) ----------------------
    def moments_to_comoments(self, *, mom, mom_dims_out=("xmom", "umom")) -> "CentralMomentsData":
        if self.mom_ndim != 1:
            raise ValueError("only for mom_ndim = 1")
        return CentralMomentsData(
            moments_to_comoments_device(self._dev, mom), mom_ndim=2, dims=self.val_dims + tuple(mom_dims_out)
        )


def moments_to_comoments_device(m: torch.Tensor, mom) -> torch.Tensor:
    """1-D states [..., M] -> comoment states [..., 2, K] with x == u:
    out[0, j] = m[j], out[1, j] = m[j + 1]  (out[1, 0] = <u>)."""
    M = m.shape[-1]
    m0, m1 = mom
    if m0 != 1:
        raise NotImplementedError("only mom = (1, n)")
    K = (M - 1) if m1 == -1 else m1 + 1
    if K + 1 > M:
        raise ValueError("not enough moments")
    out = torch.empty(*m.shape[:-1], 2, K, dtype=m.dtype, device=m.device)
    out[..., 0, :] = m[..., :K]
    out[..., 1, :] = m[..., 1 : K + 1]
    return out


# ---------------------------------------------------------------------------
# module-level functions
# ---------------------------------------------------------------------------
def wrap(data, mom_ndim=2, mom_dims=None, **kw) -> CentralMomentsData:
    return CentralMomentsData(data, mom_ndim=mom_ndim, mom_dims=mom_dims)


def _mom_axes(lab: DataArray, mom_ndim, mom_dims):
    if mom_dims is None:
        return tuple(range(lab.ndim - mom_ndim, lab.ndim))
    mom_dims = (mom_dims,) if isinstance(mom_dims, str) else tuple(mom_dims)
    return tuple(lab.dims.index(d) for d in mom_dims)


def select_moment(data, name: str, *, mom_ndim=2, mom_dims=None) -> DataArray:
    """cmomy.select_moment for the names thermoextrap uses (data.py:486-492, 857-909)."""
    lab = as_labelled(data.obj if isinstance(data, CentralMomentsData) else data)
    ax = _mom_axes(lab, mom_ndim, mom_dims)
    md = tuple(lab.dims[a] for a in ax)
    if mom_ndim == 2:
        xd, ud = md
        if name == "xmom_0":
            return lab.isel({xd: 0}, drop=True)
        if name == "xmom_1":
            return lab.isel({xd: 1}, drop=True)
        if name == "xave":
            return lab.isel({xd: 1, ud: 0}, drop=True)
        if name == "yave":
            return lab.isel({xd: 0, ud: 1}, drop=True)
        if name == "weight":
            return lab.isel({xd: 0, ud: 0}, drop=True)
    else:
        (ud,) = md
        if name == "weight":
            return lab.isel({ud: 0}, drop=True)
        if name in ("ave", "xave"):
            return lab.isel({ud: 1}, drop=True)
    raise ValueError(f"unknown moment name {name!r} for mom_ndim={mom_ndim}")


def assign_moment(data, *, weight=None, mom_dims=None, mom_ndim=None, copy=True, **kw) -> DataArray:
    """cmomy.assign_moment(..., weight=) (data.py:1449-1460)."""
    lab = as_labelled(data)
    mom_dims = (mom_dims,) if isinstance(mom_dims, str) else tuple(mom_dims)
    out = lab.copy() if copy else lab
    if weight is not None:
        order = [d for d in out.dims if d not in mom_dims] + list(mom_dims)
        t = out.transpose(*order)
        w = as_labelled(weight) if is_labelled(weight) else weight
        idx = (Ellipsis,) + (0,) * len(mom_dims)
        if isinstance(w, DataArray):
            lead = [d for d in order if d not in mom_dims]
            t.values[idx] = np.broadcast_to(np.asarray((w * DataArray(np.ones([t.sizes[d] for d in lead]), lead)).transpose(*lead).values), t.values[idx].shape)
        else:
            t.values[idx] = w
        out = t
    return out


def _batch_over(x_t, x_dims, u_t, u_dims, red_dim):
    """yield (batch index tuple over u's extra dims, x slice, u slice) with the
    reduction dim handled by the caller."""
    u_extra = [d for d in u_dims if d != red_dim]
    for d in u_extra:
        if d not in x_dims:
            raise ValueError(f"dimension {d!r} of uv missing from xv {x_dims}")
    if not u_extra:
        yield (), x_t, x_dims, u_t
        return
    sizes = [u_t.shape[u_dims.index(d)] for d in u_extra]
    for flat in np.ndindex(*sizes):
        xs, us = x_t, u_t
        xd, ud = list(x_dims), list(u_dims)
        for d, i in zip(u_extra, flat):
            xs = xs.select(xd.index(d), i)
            xd.remove(d)
            us = us.select(ud.index(d), i)
            ud.remove(d)
        yield flat, xs, tuple(xd), us


def wrap_reduce_vals(x, *y, mom, weight=None, axis=MISSING, dim=MISSING, mom_dims=None, **kw) -> CentralMomentsData:
    """cmomy.wrap_reduce_vals (data.py:485-489, 528-532, 1183-1203, 1632-1640).

    mom = int      : 1-D central moments of `x` along `dim`.
    mom = (1, n)   : comoments of `x` against `y[0]` (= uv) along `dim`.
    """
    del kw
    xt, xdims = _dev_and_dims(x)
    red = _resolve_dim(xdims, dim, axis)
    N = xt.shape[xdims.index(red)]
    wt = None
    if weight is not None:
        wt, wdims = _dev_and_dims(weight) if (is_labelled(weight) or isinstance(weight, DeviceDataArray)) else (
            engine.to_device(np.asarray(weight)), (red,))
        if wdims != (red,):
            raise NotImplementedError("weight must be 1-D along the reduction dim")
    if isinstance(mom, int):
        mdim = mom_dims if isinstance(mom_dims, str) else ((mom_dims or ("mom_0",))[0])
        others = [d for d in xdims if d != red]
        perm = [xdims.index(d) for d in others] + [xdims.index(red)]
        t = xt.permute(perm)
        oshape = t.shape[:-1]
        t2 = t.reshape(-1, N) if t.dim() > 1 else t
        out = engine.reduce_vals_1d(t2 if t2.dim() == 2 else t2.unsqueeze(0), mom, w=wt)
        out = out.reshape(*oshape, mom + 1)
        return CentralMomentsData(out, mom_ndim=1, dims=tuple(others) + (mdim,))

    if len(y) != 1 or tuple(mom)[0] != 1:
        raise NotImplementedError("comoments need exactly one y array and mom = (1, n)")
    order = int(mom[1])
    ut, udims = _dev_and_dims(y[0])
    if red not in udims:
        raise ValueError(f"dimension {red!r} not found in uv dimensions {udims}")
    md = tuple(mom_dims) if mom_dims is not None else ("mom_0", "mom_1")
    u_extra = [d for d in udims if d != red]
    x_others = [d for d in xdims if d != red]
    results = {}
    for flat, xs, xd, us in _batch_over(xt, xdims, ut, udims, red):
        cols = [d for d in xd if d != red]
        ax = xd.index(red)
        # (N, C) view without copying when the layout allows it
        x2 = xs.movedim(ax, 0)
        cshape = x2.shape[1:]
        x2 = x2.reshape(N, -1) if x2.dim() > 1 else x2
        st = engine.reduce_vals(x2, us.contiguous(), order, w=wt)
        results[flat] = (st.reshape(*cshape, 2, order + 1), cols)
    if not u_extra:
        st, cols = results[()]
        return CentralMomentsData(st, mom_ndim=2, dims=tuple(cols) + md)
    sizes = [ut.shape[udims.index(d)] for d in u_extra]
    first, cols = results[(0,) * len(u_extra)]
    stacked = torch.stack([results[f][0] for f in np.ndindex(*sizes)]).reshape(*sizes, *first.shape)
    cur = tuple(u_extra) + tuple(cols) + md
    target = tuple(d for d in x_others) + md  # cmomy keeps x's dim order
    return CentralMomentsData(stacked.permute([cur.index(d) for d in target]).contiguous(), mom_ndim=2, dims=target)


def wrap_resample_vals(x, *y, mom, sampler, weight=None, axis=MISSING, dim=MISSING, mom_dims=None,
                       rep_dim="rep", parallel=None, _prep=None, **kw) -> CentralMomentsData:
    """cmomy.wrap_resample_vals (data.py:1354-1366, 1803-1810): dims of the result
    are x's with `dim` replaced by `rep_dim`, plus the moment dims.  `_prep` (extension): the owning data
    object's engine.ResamplePrep -- the int8 path's pre-pass tables are then computed once per data object."""
    del parallel, kw
    if len(y) != 1 or tuple(mom)[0] != 1:
        raise NotImplementedError("comoments need exactly one y array and mom = (1, n)")
    order = int(mom[1])
    xt, xdims = _dev_and_dims(x)
    ut, udims = _dev_and_dims(y[0])
    red = _resolve_dim(xdims, dim, axis)
    if udims != (red,):
        raise NotImplementedError("uv must be 1-D along the resampled dim")
    N = xt.shape[xdims.index(red)]
    sampler = factory_sampler(sampler, data=x, dim=red, rep_dim=rep_dim)
    if sampler.ndat != N:
        raise ValueError(f"sampler.ndat={sampler.ndat} must equal {N}")
    wt = None
    if weight is not None:
        wt, _ = _dev_and_dims(weight) if (is_labelled(weight) or isinstance(weight, DeviceDataArray)) else (
            engine.to_device(np.asarray(weight)), None)
    md = tuple(mom_dims) if mom_dims is not None else ("mom_0", "mom_1")
    others = [d for d in xdims if d != red]
    x2 = xt.movedim(xdims.index(red), 0)
    cshape = x2.shape[1:]
    x2 = x2.reshape(N, -1) if x2.dim() > 1 else x2
    if x2.dim() == 2 and x2.stride(1) != 1:
        x2 = x2.contiguous()
    if sampler.is_device:
        # the pre-pass block is keyed on the arrays of the data object (xt, ut, wt), not on the views / copies made here
        st = engine.resample_vals(x2, ut.contiguous(), order, sampler=sampler.device_sampler, w=wt, prep=_prep,
                                  prep_src=(xt, ut, wt))
    else:
        st = engine.resample_vals(x2, ut.contiguous(), order, freq=sampler.freq_device(), w=wt)
    st = st.reshape(sampler.nrep, *cshape, 2, order + 1)
    cur = (rep_dim, *others, *md)
    target = tuple(rep_dim if d == red else d for d in xdims) + md
    return CentralMomentsData(st.permute([cur.index(d) for d in target]).contiguous(), mom_ndim=2, dims=target)


# ---------------------------------------------------------------------------
# convert (cmomy.convert)
# ---------------------------------------------------------------------------
def _moments_type(values, *, mom_ndim=2, mom_dims=None, to="central", **kw):
    """cmomy.convert.moments_type(raw, to="central") (data.py:1109-1115)."""
    del kw
    lab = as_labelled(values)
    ax = _mom_axes(lab, mom_ndim, mom_dims)
    md = tuple(lab.dims[a] for a in ax)
    order = [d for d in lab.dims if d not in md] + list(md)
    t = engine.to_device(lab.transpose(*order).values)
    to_central = to == "central"
    out = engine.convert_cov(t, to_central) if mom_ndim == 2 else engine.convert_1d(t, to_central)
    res = DataArray(out.cpu().numpy(), order)
    res._inherit(lab._coords)
    return res.transpose(*lab.dims)


def _comoments_to_moments(values, *, mom_dims=None, mom_dims_out="umom", **kw):
    """[..., 2, K] (x == u) -> [..., K + 1]: m[j] = c[0, j], m[K] = c[1, K-1] (data.py:866-872, 899-902)."""
    del kw
    lab = as_labelled(values)
    ax = _mom_axes(lab, 2, mom_dims)
    xd, ud = lab.dims[ax[0]], lab.dims[ax[1]]
    lead = [d for d in lab.dims if d not in (xd, ud)]
    v = lab.transpose(*lead, xd, ud).values
    K = v.shape[-1]
    out = np.empty(v.shape[:-2] + (K + 1,))
    out[..., :K] = v[..., 0, :]
    out[..., K] = v[..., 1, K - 1]
    mo = mom_dims_out if isinstance(mom_dims_out, str) else mom_dims_out[0]
    res = DataArray(out, (*lead, mo))
    res._inherit(lab._coords)
    return res


def _moments_to_comoments(values, *, mom, mom_dims=None, mom_dims_out=("xmom", "umom"), **kw):
    del kw
    lab = as_labelled(values)
    ax = _mom_axes(lab, 1, mom_dims)
    ud = lab.dims[ax[0]]
    lead = [d for d in lab.dims if d != ud]
    t = torch.as_tensor(lab.transpose(*lead, ud).values)
    out = moments_to_comoments_device(t, mom).numpy()
    return DataArray(out, (*lead, *mom_dims_out))


convert = SimpleNamespace(
    moments_type=_moments_type,
    comoments_to_moments=_comoments_to_moments,
    moments_to_comoments=_moments_to_comoments,
)
