"""Device-level calls into libtxmom: torch CUDA tensors in, torch CUDA tensors out.

Thin layer over the C ABI (include/txmom.h): allocates outputs and scratch with
torch, passes raw pointers and the current torch stream, checks status codes.
The labelled-array API that mirrors the reference sits on top (moments.py,
data.py); nothing here knows about dims or names.
"""

from __future__ import annotations

import contextlib
import ctypes as ct

import numpy as np
import torch

from . import _lib
from ._lib import ResampleOpts, SamplerSpec, check

F64 = torch.float64
_last_info: dict[tuple[int, int], torch.Tensor] = {}
_ws_cache: dict[tuple[int, int, str], torch.Tensor] = {}


def _L():
    if not _lib.gpu_ready():
        _lib.require_gpu()
    return _lib.load()


def _stream():
    return ct.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ct.c_void_p(t.data_ptr()) if t is not None else None


def workspace(nbytes: int, tag: str = "main") -> torch.Tensor:
    """Grow-only scratch buffer per (device, stream, tag): calls issued on different torch streams never
    share scratch, and a buffer is only ever used (and, when it grows, released) on the stream it was
    allocated on, which is what torch's caching allocator assumes."""
    key = (torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream, tag)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = None
        _ws_cache.pop(key, None)
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device="cuda")
        _ws_cache[key] = buf
    return buf


_const_cache: "dict" = {}


def const_tensor(values, dtype=F64) -> torch.Tensor:
    """A small constant (a pointer table, the 1 / n! factors, a callback's scalars) as a device tensor, uploaded ONCE per
    (values, dtype, device, stream) from pinned memory without blocking the host, then served from a cache.
    ``torch.tensor(list, device="cuda")`` is a pageable host-to-device copy: the host waits until everything queued on the stream
    before it has run -- in the middle of a step (between the bootstrap launch and the derivative evaluation) that serialises host
    and device and leaves the device idle while the host issues the step's small launches (config 5: 0.4 ms of a 6.2 ms step)."""
    key = (tuple(values), dtype, torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)
    hit = _const_cache.get(key)
    if hit is None:
        if len(_const_cache) >= 512:          # (bounded: pointer tables of long-lived programs repeat; anything else is transient)
            _const_cache.pop(next(iter(_const_cache)))
        host = torch.tensor(list(values), dtype=dtype).pin_memory()
        hit = _const_cache[key] = (host.to("cuda", non_blocking=True), host)   # (the pinned source lives as long as the entry)
    return hit[0]


def to_device(a, dtype=F64) -> torch.Tensor:
    if isinstance(a, torch.Tensor):
        return a.to(device="cuda", dtype=dtype)
    return torch.as_tensor(np.asarray(a), dtype=dtype).to("cuda")


def _check_f64_cuda(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == F64):
        raise TypeError(f"{name} must be a float64 CUDA tensor")


# ---------------------------------------------------------------------------
def reduce_vals(x: torch.Tensor, u: torch.Tensor, order: int, w: torch.Tensor | None = None) -> torch.Tensor:
    """x: (N, C) with either stride(1) == 1 or stride(0) == 1 (a transposed view
    of a (C, N) array), or (N,).  Returns (C, 2, K) (or (2, K))."""
    L = _L()
    _check_f64_cuda(x, "x")
    _check_f64_cuda(u, "u")
    squeeze = x.dim() == 1
    x2 = x.unsqueeze(1) if squeeze else x
    if x2.dim() != 2:
        raise ValueError("x must be (N,) or (N, C)")
    N, C = x2.shape
    if u.shape != (N,):
        raise ValueError(f"u must have shape ({N},), got {tuple(u.shape)}")
    u = u.contiguous()
    if w is not None:
        _check_f64_cuda(w, "w")
        if w.shape != (N,):
            raise ValueError("w must have shape (N,)")
        w = w.contiguous()
    if C == 1:
        x2 = x2.contiguous()
        ls, lc = 1, 1  # a single contiguous series
    elif x2.stride(1) == 1 and (N == 1 or x2.stride(0) >= C):
        ls, lc = (x2.stride(0) if N > 1 else C), 1  # (rec, val) row-major, pitch ls
    elif x2.stride(0) == 1 and (C == 1 or x2.stride(1) >= N):
        ls, lc = 1, x2.stride(1)  # (val, rec): every column contiguous along samples
    else:
        x2 = x2.contiguous()
        ls, lc = C, 1
    out = torch.empty((C, 2, order + 1), dtype=F64, device="cuda")
    nws = L.txm_reduce_vals_ws_bytes(N, C, order)
    ws = workspace(nws)
    check(
        L.txm_reduce_vals(_ptr(x2), ls, lc, _ptr(u), _ptr(w), N, C, order, _ptr(out), _ptr(ws), ws.numel(), _stream()),
        "txm_reduce_vals",
    )
    return out[0] if squeeze else out


def _rowmajor_operands(x, u, w):
    """(x2, ls, lc, u, w, N, C, squeeze) for the sample-matrix entry points: (N, C) or (N,), either layout of reduce_vals."""
    _check_f64_cuda(x, "x")
    _check_f64_cuda(u, "u")
    squeeze = x.dim() == 1
    x2 = x.unsqueeze(1) if squeeze else x
    if x2.dim() != 2:
        raise ValueError("x must be (N,) or (N, C)")
    N, C = x2.shape
    if u.shape != (N,):
        raise ValueError(f"u must have shape ({N},), got {tuple(u.shape)}")
    u = u.contiguous()
    if w is not None:
        _check_f64_cuda(w, "w")
        if w.shape != (N,):
            raise ValueError("w must have shape (N,)")
        w = w.contiguous()
    if C == 1:
        x2 = x2.contiguous()
        ls, lc = 1, 1
    elif x2.stride(1) == 1 and (N == 1 or x2.stride(0) >= C):
        ls, lc = (x2.stride(0) if N > 1 else C), 1
    elif x2.stride(0) == 1 and (C == 1 or x2.stride(1) >= N):
        ls, lc = 1, x2.stride(1)
    else:
        x2 = x2.contiguous()
        ls, lc = C, 1
    return x2, ls, lc, u, w, N, C, squeeze


def reduce_pivot(x: torch.Tensor, u: torch.Tensor) -> torch.Tensor:
    """The library's pivot estimate {pivot_u, pivot_x[...]} for a (shard of a) sample matrix: (1 + C,) (txm_reduce_vals_pivot)."""
    L = _L()
    x2, ls, lc, u, _, N, C, _ = _rowmajor_operands(x, u, None)
    piv = torch.empty(1 + C, dtype=F64, device="cuda")
    check(L.txm_reduce_vals_pivot(_ptr(x2), ls, lc, _ptr(u), N, C, _ptr(piv), _stream()), "txm_reduce_vals_pivot")
    return piv


def reduce_sums(x: torch.Tensor, u: torch.Tensor, order: int, pivot: torch.Tensor, w: torch.Tensor | None = None) -> torch.Tensor:
    """Weight-scaled power sums of the samples about ``pivot``: (C, 2, K) -- sums about one pivot add like the samples
    (txm_reduce_vals_sums): shards of a sample-sharded reduce, chunks of a stream."""
    L = _L()
    x2, ls, lc, u, w, N, C, _ = _rowmajor_operands(x, u, w)
    _check_f64_cuda(pivot, "pivot")
    pivot = pivot.contiguous()
    if pivot.numel() != 1 + C:
        raise ValueError("pivot must have 1 + C entries")
    out = torch.empty((C, 2, order + 1), dtype=F64, device="cuda")
    ws = workspace(L.txm_reduce_vals_ws_bytes(N, C, order))
    check(L.txm_reduce_vals_sums(_ptr(x2), ls, lc, _ptr(u), _ptr(w), N, C, order, _ptr(pivot), _ptr(out), _ptr(ws), ws.numel(),
                                 _stream()), "txm_reduce_vals_sums")
    return out


def sums_to_state(sums: torch.Tensor, pivot: torch.Tensor) -> torch.Tensor:
    """(n, C, 2, K) stacks of sums about ``pivot`` (or one (C, 2, K)) -> the cmomy state (C, 2, K) of all their samples,
    the stacks added in index order (txm_sums_to_state)."""
    L = _L()
    _check_f64_cuda(sums, "sums")
    s = sums.contiguous()
    if s.dim() == 3:
        s = s.unsqueeze(0)
    n, C, two, K = s.shape
    if two != 2 or pivot.numel() != 1 + C:
        raise ValueError("sums must be (n, C, 2, K) with a (1 + C,) pivot")
    out = torch.empty((C, 2, K), dtype=F64, device="cuda")
    check(L.txm_sums_to_state(_ptr(s), n, _ptr(pivot.contiguous()), C, K - 1, _ptr(out), _stream()), "txm_sums_to_state")
    return out


def push_vals(state: torch.Tensor, x: torch.Tensor, u: torch.Tensor, w: torch.Tensor | None = None) -> torch.Tensor:
    """Accumulate the samples (x, u[, w]) into ``state`` (C, 2, K) IN PLACE (cmomy push_vals; zeros = the empty
    accumulator) and return it (txm_push_vals)."""
    L = _L()
    _check_f64_cuda(state, "state")
    x2, ls, lc, u, w, N, C, squeeze = _rowmajor_operands(x, u, w)
    st = state.unsqueeze(0) if (squeeze and state.dim() == 2) else state
    if st.dim() != 3 or st.shape[0] != C or st.shape[1] != 2 or not st.is_contiguous():
        raise ValueError(f"state must be a contiguous ({C}, 2, K) tensor, got {tuple(state.shape)}")
    order = st.shape[2] - 1
    ws = workspace(L.txm_push_vals_ws_bytes(N, C, order))
    check(L.txm_push_vals(_ptr(st), _ptr(x2), ls, lc, _ptr(u), _ptr(w), N, C, order, _ptr(ws), ws.numel(), _stream()),
          "txm_push_vals")
    return state


def reduce_vals_1d(u: torch.Tensor, mom: int, w: torch.Tensor | None = None) -> torch.Tensor:
    """u: (R, N) rows contiguous, or (N,).  Returns (R, mom+1) / (mom+1,)."""
    L = _L()
    _check_f64_cuda(u, "u")
    squeeze = u.dim() == 1
    u2 = u.unsqueeze(0) if squeeze else u
    if u2.stride(1) != 1:
        u2 = u2.contiguous()
    R, N = u2.shape
    if w is not None:
        _check_f64_cuda(w, "w")
        w = w.contiguous()
    M = mom + 1
    out = torch.empty((R, M), dtype=F64, device="cuda")
    ws = workspace(L.txm_reduce_vals_1d_ws_bytes(N, R, M))
    check(
        L.txm_reduce_vals_1d(_ptr(u2), u2.stride(0) if R > 1 else N, 1, _ptr(w), N, R, M, _ptr(out), _ptr(ws),
                             ws.numel(), _stream()),
        "txm_reduce_vals_1d",
    )
    return out[0] if squeeze else out


def indices_to_freq(indices: torch.Tensor, ndat: int | None = None) -> torch.Tensor:
    L = _L()
    idx = indices.to(device="cuda", dtype=torch.int64).contiguous()
    nrep, nsamp = idx.shape
    ndat = nsamp if ndat is None else int(ndat)
    freq = torch.empty((nrep, ndat), dtype=torch.int64, device="cuda")
    ws = workspace(L.txm_indices_to_freq_ws_bytes(), "indices")
    check(L.txm_indices_to_freq(_ptr(idx), nrep, nsamp, ndat, _ptr(freq), _ptr(ws), ws.numel(), _stream()),
          "txm_indices_to_freq")
    return freq


class DeviceSampler:
    """Counter-based exact multinomial sampler (include/txmom.h txm_sampler_*).

    Holds only the per-(replicate, tile) draw counts (uint32, nrep x ceil(ndat/1024));
    the per-sample counts are regenerated inside the bootstrap kernel.

    ``rep0``: row r of this sampler is replicate ``rep0 + r`` of the stream of ``seed`` -- a replicate's draws
    depend on (seed, stream replicate, tile) only, so ``DeviceSampler(seed, b - a, ndat, rep0=a)`` is rows
    ``a:b`` of ``DeviceSampler(seed, n, ndat)`` bit for bit.  That is how replicate slabs and state shards of a
    multi-GPU run reproduce the one-GPU result exactly (distributed.py).
    """

    def __init__(self, seed: int, nrep: int, ndat: int, nsamp: int = 0, rep0: int = 0):
        L = _L()
        self.spec = SamplerSpec(seed=0, nrep=int(nrep), ndat=int(ndat), nsamp=int(nsamp), rep0=int(rep0))
        self.ntiles = int(L.txm_sampler_ntiles(ndat))
        self.counts = torch.empty((nrep, self.ntiles), dtype=torch.int32, device="cuda")  # bit pattern of uint32
        self._nws = L.txm_sampler_counts_ws_bytes(ct.byref(self.spec))
        if self._nws == 0:
            raise _lib.TxmError(f"sampler spec rejected: {_lib.last_error()}")
        self.draw(seed)

    def draw(self, seed: int) -> "DeviceSampler":
        """(Re)draw the tile counts for a new seed, reusing the buffers."""
        L = _L()
        self.spec.seed = int(seed) & (2**64 - 1)
        ws = workspace(self._nws, "sampler")
        check(
            L.txm_sampler_tile_counts(ct.byref(self.spec), _ptr(self.counts), _ptr(ws), ws.numel(), _stream()),
            "txm_sampler_tile_counts",
        )
        return self

    @property
    def seed(self):
        return self.spec.seed

    @property
    def nrep(self):
        return self.spec.nrep

    @property
    def ndat(self):
        return self.spec.ndat

    @property
    def rep0(self):
        return self.spec.rep0

    def rows(self, a: int, b: int) -> "DeviceSampler":
        """Replicates [a, b) of this sampler as a sampler of their own (a view: the same count rows, stream offset rep0 + a)."""
        if not 0 <= a < b <= self.spec.nrep:
            raise ValueError(f"rows [{a}, {b}) outside [0, {self.spec.nrep})")
        v = object.__new__(DeviceSampler)
        v.spec = SamplerSpec(seed=self.spec.seed, nrep=b - a, ndat=self.spec.ndat, nsamp=self.spec.nsamp, rep0=self.spec.rep0 + a)
        v.ntiles, v.counts, v._nws = self.ntiles, self.counts[a:b], self._nws
        return v

    def freq(self) -> torch.Tensor:
        L = _L()
        out = torch.empty((self.spec.nrep, self.spec.ndat), dtype=torch.int64, device="cuda")
        check(L.txm_sampler_freq(ct.byref(self.spec), _ptr(self.counts), _ptr(out), _stream()), "txm_sampler_freq")
        return out


class ResamplePrep:
    """Persistent pre-pass block of the int8 bootstrap path for ONE set of sample arrays
    (txm_resample_opts.prep): pivot, per-window scale table, guard flags and the FP64 fallback list depend on
    (x, u, w, pivot, N, C, nrep, order) only, so a bootstrap loop over the same data computes them once.  The
    reference caches per data object the same way (data.py:285, 844-942).

    The key is built from the CALLER's tensors (storage pointer, torch version counter, shape, strides) -- not from the
    contiguous temporaries a call may have to make -- plus (N, C, order) and the KERNEL WORD of the call
    (txm_resample_kernel: fused / count-table kernel, and whether it carries a second matrix -- a block holds y's tables only
    then); the block keeps strong references to the tensors, so their storage cannot be recycled under the key; the
    temporaries themselves are kept here and reused with the tables.  An in-place edit of a sample array through torch,
    another shape, or a replicate count that the dispatch rule sends to another kernel therefore invalidates the block (the
    replicate slabs of one call and calls with other counts on the same kernel share it); so does ``new_like`` on the
    owning data object (a fresh cache).  What torch cannot see -- another library writing the
    samples through a raw pointer -- does not bump a version counter: call ``invalidate()`` after such a write (the C
    ABI has no such state: ``prep_valid`` there is the caller's own statement)."""

    def __init__(self):
        self.buf: torch.Tensor | None = None
        self.key = None
        self.refs = None     # the caller's tensors the key describes
        self.tensors = None  # what the kernels were given: (x2, u, w, pivot, y2), contiguous copies where needed
        self.hits = 0
        self.misses = 0

    def lookup(self, key):
        """The kernel operands of the last committed call if ``key`` still describes it, else None."""
        return self.tensors if (key is not None and self.key == key and self.buf is not None) else None

    def bind(self, key, nbytes: int) -> tuple[torch.Tensor, bool]:
        valid = self.key == key and self.buf is not None and self.buf.numel() >= nbytes
        if not valid:
            if self.buf is None or self.buf.numel() < nbytes:
                self.buf = torch.empty(int(nbytes), dtype=torch.uint8, device="cuda")
            self.key = None  # set by commit() once the call that fills the block has been issued
            self.refs = self.tensors = None
            self.misses += 1
        else:
            self.hits += 1
        return self.buf, valid

    def commit(self, key, refs=None, tensors=None):
        self.key = key
        self.refs, self.tensors = refs, tensors

    def invalidate(self):
        """Forget the tables (and the cached operand copies): the next call recomputes them.  Needed only after a write
        to the sample arrays that torch's version counters do not see."""
        self.key = None
        self.refs = self.tensors = None


def _tkey(t):
    return None if t is None else (t.data_ptr(), t._version, tuple(t.shape), tuple(t.stride()))


# Workspace of ONE library call of the device-sampler bootstrap (partial sums per scaling window, and the int8 path's count
# table: one byte per replicate and sample) grows with nrep -- 3.9 GB + 102 GB at N = 1e8, nrep = 1000; ten times that at
# nrep = 1e4.  resample_vals bounds it by bootstrapping the replicates in slabs of whole 128-replicate groups: rows [a, b) of
# a call equal the (b - a)-replicate call at rep0 = a BIT FOR BIT (txm_sampler_spec.rep0), so the slabs ARE the rows of the
# unslabbed result.  None: 45 % of the device's memory, at most 80 % of what is free at the first call.
WORKSPACE_BUDGET_BYTES: int | None = None
_budget_default: dict[int, int] = {}


def workspace_budget() -> int:
    if WORKSPACE_BUDGET_BYTES is not None:
        return int(WORKSPACE_BUDGET_BYTES)
    dev = torch.cuda.current_device()
    if dev not in _budget_default:
        free, total = torch.cuda.mem_get_info(dev)
        _budget_default[dev] = int(min(0.45 * total, 0.8 * free))
    return _budget_default[dev]


def _slab_size(L, N, C, nrep, order, path, has_y) -> int:
    """Replicates per library call: all of them when the call's workspace fits the budget, else the largest multiple of 128
    that does (at least 128)."""
    need = lambda n: L.txm_resample_vals_ws_bytes_opts(N, C, n, order, path, int(has_y)) + (L.txm_resample_y_ws_bytes(N, C, n) if has_y else 0)  # noqa: E731
    budget = workspace_budget()
    if need(nrep) <= budget or nrep <= 128:
        return nrep
    lo, hi = 1, (nrep + 127) // 128  # groups of 128; need() grows with n
    while lo < hi:
        mid = (lo + hi + 1) // 2
        if need(min(mid * 128, nrep)) <= budget:
            lo = mid
        else:
            hi = mid - 1
    return min(lo * 128, nrep)


def _call_path(path) -> int:
    """The kernel one device-sampler call takes: an explicit path, else the forced_path() context, else the
    library's shape rule (txm_resample_path, which honours txm_set_resample_path)."""
    eff = path if path is not None else _forced
    if eff in ("fp64", "int8", "int8_fused", "int8_table"):
        return _PATHS[eff]
    return -1


def _table_operands_ok(x2: torch.Tensor, ls: int, C: int, y: torch.Tensor | None) -> bool:
    """txm_resample_kernel's `aligned` for the operands a call will hand the library (the library's own statement of what the
    count-table kernel's DMA needs: txm_resample_operands_aligned).  A second matrix that resample_vals will have to copy
    (not row-major with unit column stride) is judged as the contiguous copy it becomes: pitch C, torch's 256-byte alignment."""
    L = _L()
    yp, ldy = None, 0
    if y is not None:
        y2 = y.unsqueeze(1) if y.dim() == 1 else y
        if y2.dim() != 2 or y2.stride(1) != 1 or (y2.shape[0] > 1 and y2.stride(0) < C):
            yp, ldy = ct.c_void_p(256), C          # (an address with torch's allocation alignment stands in for the copy's)
        else:
            yp, ldy = ct.c_void_p(y2.data_ptr()), max(y2.stride(0) if y2.shape[0] > 1 else C, C)
    return L.txm_resample_operands_aligned(ct.c_void_p(x2.data_ptr()), ls, C, yp, ldy) == 1


def resample_vals(
    x: torch.Tensor,
    u: torch.Tensor,
    order: int,
    *,
    freq: torch.Tensor | None = None,
    sampler: DeviceSampler | None = None,
    w: torch.Tensor | None = None,
    pivot: torch.Tensor | None = None,
    out: torch.Tensor | None = None,
    path: str | None = None,
    prep: ResamplePrep | None = None,
    info: torch.Tensor | None = None,
    y: torch.Tensor | None = None,
    prep_src: tuple | None = None,
    _slab: int | None = None,
):
    """(nrep, C, 2, K) bootstrap states; x is (N, C) row-major (or (N,)).

    ``path``: "fp64" / "int8" for THIS call (None: the forced_path() context, else the library's rule).
    ``prep``: a ResamplePrep kept by the caller next to the data -- the int8 path's pre-pass is then computed once.
    ``prep_src``: the tensors the block's key should describe when x / u / w are themselves temporaries of the caller
    (default: the arguments as given, before any contiguous copy made here).
    ``info``: an int64 CUDA tensor of 4 words the library fills on the stream (path, windows, windows the guard
    sent to the FP64 kernel, tables reused) -- no synchronisation.
    ``y``: a second (N, C) sample matrix; returns ``(states, ymean)`` with ymean (nrep, C) = the per-replicate
    weighted mean of y on the same draw (VolumeDataCallback's <dx/dq>, reference volume.py:121-134)."""
    L = _L()
    _check_f64_cuda(x, "x")
    _check_f64_cuda(u, "u")
    squeeze = x.dim() == 1
    x2 = x.unsqueeze(1) if squeeze else x
    if x2.dim() != 2:
        raise ValueError("x must be (N,) or (N, C)")
    N, C = x2.shape
    # the key of a caller-held pre-pass block describes the CALLER's tensors; the copies made below are kept with it
    src = tuple(prep_src) if prep_src is not None else (x, u, w, pivot, y)
    src_key = tuple(_tkey(t) for t in src)
    # the kernels want (rec, val) row-major with a row pitch >= C; anything else (transposed views,
    # broadcast rows with stride 0, overlapping pitches) is copied
    if not (x2.stride(1) == 1 or C == 1) or (N > 1 and x2.stride(0) < C):
        x2 = x2.contiguous()
    if C == 1 and x2.stride(1) != 1:
        x2 = x2.contiguous()
    ls = max(x2.stride(0) if N > 1 else C, C)
    if u.shape != (N,):
        raise ValueError(f"u must have shape ({N},), got {tuple(u.shape)}")
    u = u.contiguous()
    if w is not None:
        _check_f64_cuda(w, "w")
        if w.shape != (N,):
            raise ValueError(f"w must have shape ({N},), got {tuple(w.shape)}")
        w = w.contiguous()
    if (freq is None) == (sampler is None):
        raise ValueError("give exactly one of freq= or sampler=")
    if freq is not None:
        freq = freq.to(device="cuda", dtype=torch.int64).contiguous()
        if freq.dim() != 2 or freq.shape[1] != N:
            raise ValueError(f"freq must be (nrep, {N}), got {tuple(freq.shape)}")
        nrep = freq.shape[0]
        spec_p, counts_p = None, None
    else:
        if sampler.ndat != N:
            raise ValueError(f"sampler.ndat={sampler.ndat} must equal N={N}")
        nrep = sampler.nrep
        spec_p, counts_p = ct.byref(sampler.spec), _ptr(sampler.counts)
    if pivot is not None:
        _check_f64_cuda(pivot, "pivot")
        pivot = pivot.contiguous()
        if pivot.numel() != 1 + C:
            raise ValueError("pivot must have 1 + C entries")
    if out is None:
        out = torch.empty((nrep, C, 2, order + 1), dtype=F64, device="cuda")
    else:
        _check_f64_cuda(out, "out")
        if tuple(out.shape) != (nrep, C, 2, order + 1) or not out.is_contiguous():
            raise ValueError(f"out must be a contiguous ({nrep}, {C}, 2, {order + 1}) tensor, got {tuple(out.shape)}")
    opts = ResampleOpts()
    opts.path = _slab if _slab is not None else _call_path(path)
    kern = -1
    if sampler is not None:
        # the contraction kernel this call runs (FP64 / int8 fused / int8 count table) and whether it carries y: decided ONCE,
        # for the whole call -- the rule looks at nrep, and neither a replicate slab nor a later call on the same pre-pass block
        # may fall on the other side of one of its thresholds (a block holds y's tables only when the kernel carries y)
        kern = L.txm_resample_kernel(N, C, nrep, order, opts.path, int(y is not None), int(_table_operands_ok(x2, ls, C, y)))
    if sampler is not None and _slab is None:
        slab = _slab_size(L, N, C, nrep, order, kern & 0xFF, y is not None)
        if slab < nrep:
            # replicate slabs: every slab is an ordinary call on its rows of the sampler with the whole call's kernel as its path
            ym = torch.empty((nrep, C), dtype=F64, device="cuda") if y is not None else None
            for a in range(0, nrep, slab):
                b = min(nrep, a + slab)
                r = resample_vals(x, u, order, sampler=sampler.rows(a, b), w=w, pivot=pivot, out=out[a:b], prep=prep,
                                  info=info, y=y, prep_src=prep_src if prep_src is not None else (x, u, w, pivot, y),
                                  _slab=kern & 0xFF)
                if y is not None:
                    ym[a:b] = r[1] if y.dim() > 1 else r[1][:, None]
            res = out[:, 0] if squeeze else out
            return (res, (ym[:, 0] if y.dim() == 1 else ym)) if y is not None else res
    ymean = y2 = None
    if y is not None:
        _check_f64_cuda(y, "y")
        y2 = y.unsqueeze(1) if y.dim() == 1 else y
        if tuple(y2.shape) != (N, C):
            raise ValueError(f"y must have the shape of x, ({N}, {C}); got {tuple(y2.shape)}")
        if y2.stride(1) != 1 or (N > 1 and y2.stride(0) < C):
            y2 = y2.contiguous()
        ymean = torch.empty((nrep, C), dtype=F64, device="cuda")
        opts.y, opts.ldy_s, opts.out_y = y2.data_ptr(), max(y2.stride(0) if N > 1 else C, C), ymean.data_ptr()
    key = None
    if prep is not None and freq is None:
        if (kern & 0xFF) != _PATHS["fp64"]:
            # the block holds pivot, window table, guard flags and fallback list of the caller's tensors -- and the second
            # matrix's tables exactly when the call's kernel carries y (txm_resample_kernel's WITH_Y bit; narrow tail groups
            # follow it too).  So the key is the kernel word, not nrep and not the path the caller asked for: the replicate
            # slabs of one call share the block, a call with another count shares it when the rule gives it the same kernel,
            # and gets a block of its own when it does not (order 4 with y: fused below two replicate groups, table above)
            key = (src_key, N, C, order, kern)
            kept = prep.lookup(key)
            if kept is not None:  # same caller tensors, unedited: the operands of the call that filled the block
                x2, u, w, pivot, y2 = kept
                ls = max(x2.stride(0) if N > 1 else C, C)
                if y2 is not None:
                    opts.y, opts.ldy_s = y2.data_ptr(), max(y2.stride(0) if N > 1 else C, C)
            buf, valid = prep.bind(key, L.txm_resample_prep_bytes(N, C, nrep, order))
            opts.prep, opts.prep_bytes, opts.prep_valid = buf.data_ptr(), buf.numel(), int(valid)
    if info is None:  # the words resample_info() reads back: one block per (device, stream)
        ikey = (torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)
        info = _last_info.get(ikey)
        if info is None:
            info = _last_info[ikey] = torch.zeros(4, dtype=torch.int64, device="cuda")
    if not (info.is_cuda and info.dtype == torch.int64 and info.numel() >= 4 and info.is_contiguous()):
        raise TypeError("info must be a contiguous int64 CUDA tensor with >= 4 elements")
    opts.info = info.data_ptr()
    ws = workspace(L.txm_resample_vals_ws_bytes_opts(N, C, nrep, order, opts.path, int(y is not None))
                   + (L.txm_resample_y_ws_bytes(N, C, nrep) if y is not None else 0))
    check(
        L.txm_resample_vals(_ptr(x2), ls, 1, _ptr(u), _ptr(w), N, C, order, nrep, _ptr(freq), spec_p, counts_p,
                            _ptr(pivot), _ptr(out), ct.byref(opts), _ptr(ws), ws.numel(), _stream()),
        "txm_resample_vals",
    )
    if key is not None:
        prep.commit(key, refs=src, tensors=(x2, u, w, pivot, y2))
    res = out[:, 0] if squeeze else out
    if y is not None:
        return res, (ymean[:, 0] if y.dim() == 1 else ymean)
    return res


def _state_table(xs, us, ws):
    """Validate S state points of one shape and build the host pointer table of the batched entry points."""
    S = len(xs)
    if S < 1 or len(us) != S or (ws is not None and len(ws) != S):
        raise ValueError("need the same number (>= 1) of x, u (and w) tensors")
    x2 = []
    for x in xs:
        _check_f64_cuda(x, "x")
        x = x.unsqueeze(1) if x.dim() == 1 else x
        if x.dim() != 2:
            raise ValueError("every x must be (N,) or (N, C)")
        x2.append(x if x.is_contiguous() else x.contiguous())
    N, C = x2[0].shape
    if any(tuple(x.shape) != (N, C) for x in x2):
        raise ValueError("the batched path needs states of one shape (N, C)")
    u2 = []
    for u in us:
        _check_f64_cuda(u, "u")
        if u.shape != (N,):
            raise ValueError(f"every u must have shape ({N},)")
        u2.append(u.contiguous())
    w2 = None
    if ws is not None:
        w2 = []
        for w in ws:
            _check_f64_cuda(w, "w")
            if w.shape != (N,):
                raise ValueError(f"every w must have shape ({N},)")
            w2.append(w.contiguous())
    tab = (_lib.StatePtrs * S)()
    for s in range(S):
        tab[s].x, tab[s].u = x2[s].data_ptr(), u2[s].data_ptr()
        tab[s].w = w2[s].data_ptr() if w2 is not None else None
    return tab, (x2, u2, w2), S, N, C


def reduce_vals_batched(xs, us, order: int, ws=None) -> torch.Tensor:
    """S state points of one shape in one launch per kernel (txm_reduce_vals_batched): (S, C, 2, K)."""
    L = _L()
    tab, keep, S, N, C = _state_table(xs, us, ws)
    out = torch.empty((S, C, 2, order + 1), dtype=F64, device="cuda")
    wsb = workspace(L.txm_reduce_vals_batched_ws_bytes(S, N, C, order))
    check(L.txm_reduce_vals_batched(tab, S, C, N, C, order, _ptr(out), _ptr(wsb), wsb.numel(), _stream()),
          "txm_reduce_vals_batched")
    del keep
    return out[:, 0] if xs[0].dim() == 1 else out


def resample_vals_batched(xs, us, order: int, *, nrep: int, sampler: DeviceSampler | None = None,
                          freq: torch.Tensor | None = None, ws=None, path: str | None = None,
                          prep: ResamplePrep | None = None) -> torch.Tensor:
    """Bootstrap of S state points of one shape in one launch per kernel (txm_resample_vals_batched_opts):
    (S, nrep, C, 2, K).  `sampler`: ONE DeviceSampler over S * nrep replicates of N samples (state s owns
    replicates s * nrep ...), or `freq`: (S * nrep, N) explicit counts.

    Narrow states (C <= 16) take the int8 path with the state on a grid axis of every kernel of it -- state s of the
    result is, bit for bit, the single call on state s with ``rep0 + s * nrep``.  ``path``: "fp64" / "int8" for THIS call
    (None: the forced_path() context, else the library's rule).  ``prep``: a ResamplePrep the caller keeps next to the
    collection -- the int8 path's pre-pass over the S states is then computed once (keyed on the callers' tensors)."""
    L = _L()
    src = (tuple(xs), tuple(us), None if ws is None else tuple(ws))
    tab, keep, S, N, C = _state_table(xs, us, ws)
    if (freq is None) == (sampler is None):
        raise ValueError("give exactly one of freq= or sampler=")
    spec_p = counts_p = None
    if freq is not None:
        freq = freq.to(device="cuda", dtype=torch.int64).contiguous()
        if tuple(freq.shape) != (S * nrep, N):
            raise ValueError(f"freq must be ({S * nrep}, {N}), got {tuple(freq.shape)}")
    else:
        if sampler.ndat != N or sampler.nrep != S * nrep:
            raise ValueError(f"the sampler must span {S} x {nrep} replicates of {N} samples")
        spec_p, counts_p = ct.byref(sampler.spec), _ptr(sampler.counts)
    out = torch.empty((S, nrep, C, 2, order + 1), dtype=F64, device="cuda")
    opts = _lib.ResampleOpts()
    opts.path = _call_path(path)
    key = None
    if prep is not None and freq is None:
        nb = L.txm_resample_batched_prep_bytes(S, N, C, nrep, order)
        # bind (and afterwards commit) the block only when THIS call runs the int8 path -- the library's own predicates, as the
        # single call does: a call that ran the FP64 kernel never fills the block, and a later int8 call would read it as valid
        takes_i8 = bool(nb) and (opts.path in (1, 2, 3) or (opts.path == -1 and L.txm_resample_batched_path(S, N, C, nrep, order) == 1))
        if takes_i8:
            key = (tuple(_tkey(t) for t in src[0]), tuple(_tkey(t) for t in src[1]),
                   None if src[2] is None else tuple(_tkey(t) for t in src[2]), S, N, C, nrep, order)
            kept = prep.lookup(key)
            if kept is not None:  # the operand copies of the call that filled the block
                tab, keep = kept
            buf, valid = prep.bind(key, nb)
            opts.prep, opts.prep_bytes, opts.prep_valid = buf.data_ptr(), buf.numel(), int(valid)
    ikey = (torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)  # the words batched_info() reads back
    info = _last_batched_info.get(ikey)
    if info is None:
        info = _last_batched_info[ikey] = torch.zeros(4, dtype=torch.int64, device="cuda")
    opts.info = info.data_ptr()
    wsb = workspace(L.txm_resample_vals_batched_ws_bytes(S, N, C, nrep, order))
    check(L.txm_resample_vals_batched_opts(tab, S, C, N, C, order, nrep, _ptr(freq), spec_p, counts_p, _ptr(out), ct.byref(opts),
                                           _ptr(wsb), wsb.numel(), _stream()), "txm_resample_vals_batched_opts")
    if key is not None:
        prep.commit(key, refs=src, tensors=(tab, keep))
    del keep
    return out[:, :, 0] if xs[0].dim() == 1 else out


_last_batched_info: dict = {}


def batched_info() -> dict:
    """What the last `resample_vals_batched` call on the current stream did (synchronises): {"path", "windows" (scaling
    windows x states), "windows_fp64" (how many of them the precision guard sent to the FP64 kernel), "prep_reused"}."""
    info = _last_batched_info.get((torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream))
    if info is None:
        return {}
    v = info.cpu().tolist()
    return {"path": "int8" if v[0] == 1 else "fp64", "windows": int(v[1]), "windows_fp64": int(v[2]), "prep_reused": bool(v[3] & 1)}


def resample_path(N: int, C: int, nrep: int, order: int) -> str:
    """Which kernel the device-sampler bootstrap takes for this shape: "fp64" or "int8" (inside a forced_path()
    context: that path wherever the int8 kernel supports the shape)."""
    L = _L()
    if _forced == "fp64":
        return "fp64"
    if _forced in ("int8", "int8_fused", "int8_table"):  # wherever the int8 kernel supports the shape: the library's own predicate
        return "int8" if L.txm_resample_i8_supported(int(N), int(C), int(nrep), int(order)) == 1 else "fp64"
    return "int8" if L.txm_resample_path(int(N), int(C), int(nrep), int(order)) == 1 else "fp64"


_PATHS = {None: -1, "auto": -1, "fp64": 0, "int8": 1, "int8_fused": 2, "int8_table": 3}
_forced: str | None = None


@contextlib.contextmanager
def forced_path(path: str | None):
    """Force the FP64 ("fp64") or the int8-sliced ("int8") bootstrap kernel wherever it applies, for the calls
    made inside the context; None / "auto" is the library's own choice.  The path travels with every call
    (txm_resample_opts.path): no process-global library state is touched.  For tests and benchmarks: the automatic
    choice plus the precision guard is what users get."""
    global _forced
    if path not in _PATHS:
        raise ValueError(f"path must be one of {sorted(k for k in _PATHS if k)} or None")
    _L()
    prev = _forced
    _forced = None if path == "auto" else path
    try:
        yield
    finally:
        _forced = prev


def resample_info(N: int | None = None, C: int | None = None, nrep: int | None = None, order: int | None = None) -> dict:
    """What the last `resample_vals` call on the current stream did: {"path", "windows", "windows_fp64",
    "prep_reused", "kernel"} -- "kernel": which contraction served the wide column groups ("int8_table" / "int8_fused" / "fp64"); "windows_fp64" counts the scaling windows (x column groups) that the precision guard of the
    int8 path handed to the FP64 kernel.  Reads the info words the library wrote on the stream
    (txm_resample_opts.info); synchronises.  The shape arguments are accepted for compatibility and ignored."""
    _L()
    info = _last_info.get((torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream))
    if info is None:
        raise RuntimeError("no resample_vals call has been made on this stream")
    v = info.cpu().tolist()
    return {"path": "int8" if v[0] == 1 else "fp64", "windows": int(v[1]), "windows_fp64": int(v[2]),
            "prep_reused": bool(v[3] & 1), "kernel": ("fp64" if v[0] != 1 else "int8_table" if v[3] & 2 else "int8_fused")}


def resample_data(data: torch.Tensor, freq: torch.Tensor | None, order: int) -> torch.Tensor:
    """data (nrec, C, 2, K); freq (nrep, nrec) or None (plain reduce) -> (nrep, C, 2, K)."""
    L = _L()
    _check_f64_cuda(data, "data")
    data = data.contiguous()
    nrec, C = data.shape[:2]
    if data.shape[2:] != (2, order + 1):
        raise ValueError("data must be (nrec, C, 2, order+1)")
    if freq is not None:
        freq = freq.to(device="cuda", dtype=torch.int64).contiguous()
        if freq.shape[1] != nrec:
            raise ValueError("freq must be (nrep, nrec)")
        nrep = freq.shape[0]
    else:
        nrep = 1
    out = torch.empty((nrep, C, 2, order + 1), dtype=F64, device="cuda")
    ws = workspace(L.txm_resample_data_ws_bytes(nrec, C, order))
    check(
        L.txm_resample_data(_ptr(data), _ptr(freq), nrec, C, nrep, order, _ptr(out), _ptr(ws), ws.numel(), _stream()),
        "txm_resample_data",
    )
    return out


def convert_cov(states: torch.Tensor, to_central: bool) -> torch.Tensor:
    L = _L()
    _check_f64_cuda(states, "states")
    s = states.contiguous()
    K = s.shape[-1]
    if s.shape[-2] != 2:
        raise ValueError("states must be (..., 2, K)")
    out = torch.empty_like(s)
    check(L.txm_convert_cov(_ptr(s), _ptr(out), s.numel() // (2 * K), K - 1, int(to_central), _stream()),
          "txm_convert_cov")
    return out


def convert_1d(states: torch.Tensor, to_central: bool) -> torch.Tensor:
    L = _L()
    _check_f64_cuda(states, "states")
    s = states.contiguous()
    M = s.shape[-1]
    out = torch.empty_like(s)
    check(L.txm_convert_1d(_ptr(s), _ptr(out), s.numel() // M, M, int(to_central), _stream()), "txm_convert_1d")
    return out


def cov_over_rep(vals: torch.Tensor) -> torch.Tensor:
    """vals (n_ord, nrep, nval) -> (nval, n_ord, n_ord), ddof = 1 (numpy.cov over replicates)."""
    L = _L()
    _check_f64_cuda(vals, "vals")
    v = vals.contiguous()
    n_ord, nrep, nval = v.shape
    out = torch.empty((nval, n_ord, n_ord), dtype=F64, device="cuda")
    check(L.txm_cov_over_rep(_ptr(v), n_ord, nrep, nval, _ptr(out), _stream()), "txm_cov_over_rep")
    return out


TAYLOR_MODES = {"sum": 0, "cumsum": 1, "terms": 2}


def predict_taylor(derivs: torch.Tensor, dalphas, mode: str = "sum") -> torch.Tensor:
    """Taylor series of a derivative table (n_ord, ...) at every dalpha: (n_alpha, ...) for ``mode="sum"``,
    (n_alpha, n_ord, ...) partial sums (``"cumsum"``) or terms (``"terms"``) -- ExtrapModel.predict in one launch."""
    L = _L()
    _check_f64_cuda(derivs, "derivs")
    if mode not in TAYLOR_MODES:
        raise ValueError(f"mode must be one of {sorted(TAYLOR_MODES)}")
    d = derivs.contiguous()
    n_ord = d.shape[0]
    if not 1 <= n_ord <= 16:
        raise ValueError("predict_taylor: 1 <= n_ord <= 16")
    M = d[0].numel()
    da = to_device(np.atleast_1d(np.asarray(dalphas, dtype=np.float64)).ravel())
    na = da.numel()
    if na < 1 or M < 1:
        raise ValueError("predict_taylor: empty input")
    shape = (na, *d.shape[1:]) if mode == "sum" else (na, n_ord, *d.shape[1:])
    out = torch.empty(shape, dtype=F64, device="cuda")
    # the kernel carries alpha on a 16-bit grid axis: more than 65535 alpha values go in slices (same launch shape)
    step = 65535
    for a0 in range(0, na, step):
        a1 = min(na, a0 + step)
        check(L.txm_predict_taylor(_ptr(d), n_ord, M, _ptr(da[a0:a1]), a1 - a0, TAYLOR_MODES[mode], _ptr(out[a0:a1]),
                                   _stream()), "txm_predict_taylor")
    return out


def perturb(x: torch.Tensor, u: torch.Tensor, dalphas, freq: torch.Tensor | None = None) -> torch.Tensor:
    """Exponentially reweighted averages for each dalpha: (n_alpha, C), or
    (nrep, n_alpha, C) with bootstrap counts ``freq`` (nrep, N).  x: (N, C) row-major or (N,)."""
    L = _L()
    _check_f64_cuda(x, "x")
    _check_f64_cuda(u, "u")
    squeeze = x.dim() == 1
    x2 = x.unsqueeze(1) if squeeze else x
    if x2.stride(1) != 1 or (x2.shape[0] > 1 and x2.stride(0) < x2.shape[1]):
        x2 = x2.contiguous()
    N, C = x2.shape
    if u.shape != (N,):
        raise ValueError(f"u must have shape ({N},), got {tuple(u.shape)}")
    u = u.contiguous()
    da = np.atleast_1d(np.asarray(dalphas, dtype=np.float64))
    nrep = 1
    if freq is not None:
        freq = freq.to(device="cuda", dtype=torch.int64).contiguous()
        if freq.dim() != 2 or freq.shape[1] != N:
            raise ValueError(f"freq must be (nrep, {N}), got {tuple(freq.shape)}")
        nrep = freq.shape[0]
    outs = []
    for a0 in range(0, len(da), 8):  # 8 perturbations per pass over the samples
        chunk = np.ascontiguousarray(da[a0 : a0 + 8])
        na = len(chunk)
        out = torch.empty((nrep, na, C), dtype=F64, device="cuda")
        ws = workspace(L.txm_perturb_ws_bytes(N, C, na, nrep))
        check(
            L.txm_perturb(_ptr(x2), max(x2.stride(0), C) if N > 1 else C, _ptr(u), N, C,
                          chunk.ctypes.data_as(ct.POINTER(ct.c_double)), na, _ptr(freq), nrep, _ptr(out), _ptr(ws),
                          ws.numel(), _stream()),
            "txm_perturb",
        )
        outs.append(out)
    res = torch.cat(outs, dim=1)
    if freq is None:
        res = res[0]
    return res[..., 0] if squeeze else res
