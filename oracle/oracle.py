"""
oracle/oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT.

numpy/ctypes front end of the CPU oracle for thermoextrap's moment hot path.

* ``liboracle.so`` (oracle/cmomy_oracle.c, oracle/philox_oracle.c) restates the
  algorithms of the third-party engine the reference delegates to
  (cmomy==0.24.0, pinned at /root/reference/uv.lock:454-455): Pebay one-pass
  weighted central comoment push, freq-weighted bootstrap, state merge,
  raw<->central conversion, indices->freq.
* The pure-numpy helpers below restate the *thermoextrap side* of the path:
  which slices of the ``[..., 2, K]`` state are ``u, xu, du, dxdu, xave``
  (/root/reference/src/thermoextrap/data.py:844-962), the ``x_is_u``
  moments<->comoments reshuffle (data.py:866-872, 899-902, 1182-1191) and the
  sampler semantics (``Generator.choice`` + bincount).

Pinned against: tests/golden/kat_notebooks.json (seeded notebook outputs of the
reference), tests/golden/fixture_legacy.npz (outputs of the reference's own
legacy oracle run in the build container, see tests/golden/make_golden.py),
tests/golden/lnpi_sample_data.json (the reference's only committed golden file).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.
"""

from __future__ import annotations

import ctypes as ct
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "_build" / "liboracle.so"
_lib = None

c_dp = ct.POINTER(ct.c_double)
c_ip = ct.POINTER(ct.c_int64)
c_u32p = ct.POINTER(ct.c_uint32)


def build(force: bool = False) -> Path:
    """Compile liboracle.so with gcc (idempotent)."""
    srcs = [_HERE / "cmomy_oracle.c", _HERE / "philox_oracle.c"]
    if (
        force
        or not _LIB_PATH.exists()
        or any(s.stat().st_mtime > _LIB_PATH.stat().st_mtime for s in srcs)
    ):
        subprocess.run(["make", "-C", str(_HERE), "-s", "-B"], check=True)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ct.CDLL(str(_LIB_PATH))
        _lib.orc_sampler_ntiles.restype = ct.c_int64
        _lib.orc_sampler_ntiles.argtypes = [ct.c_int64]
    return _lib


def _d(a):
    return a.ctypes.data_as(c_dp)


def _i(a):
    return a.ctypes.data_as(c_ip)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _strides(x):
    """x as (N, C) view -> element strides (ldx_s, ldx_c)."""
    return x.strides[0] // 8, x.strides[1] // 8


# ---------------------------------------------------------------------------
# cmomy restatement
# ---------------------------------------------------------------------------
def reduce_vals(x, u, order, w=None, nthreads=1):
    """cmomy.wrap_reduce_vals(x, u, weight=w, mom=(1, order)) along axis 0.

    x: (N,) or (N, C); returns (C, 2, K) (or (2, K) for 1-D x).
    """
    x = np.asarray(x, dtype=np.float64)
    squeeze = x.ndim == 1
    x2 = x.reshape(x.shape[0], -1)
    if x2.strides[0] % 8 or x2.strides[1] % 8:
        x2 = np.ascontiguousarray(x2)
    u = _f64(u)
    N, C = x2.shape
    K = order + 1
    out = np.zeros((C, 2, K))
    wp = _d(_f64(w)) if w is not None else None
    ls, lc = _strides(x2)
    lib().orc_reduce_vals(
        _d(x2), ct.c_int64(ls), ct.c_int64(lc), _d(u), wp,
        ct.c_int64(N), ct.c_int64(C), ct.c_int(order), _d(out), ct.c_int(nthreads),
    )
    return out[0] if squeeze else out


def reduce_vals_1d(u, mom, w=None, nthreads=1):
    """cmomy.wrap_reduce_vals(u, weight=w, mom=mom) along the last axis. u: (R, N) or (N,)."""
    u = _f64(u)
    squeeze = u.ndim == 1
    u2 = u.reshape(-1, u.shape[-1])
    R, N = u2.shape
    M = mom + 1
    out = np.zeros((R, M))
    wp = _d(_f64(w)) if w is not None else None
    lib().orc_reduce_vals_1d(
        _d(u2), ct.c_int64(N), ct.c_int64(1), wp, ct.c_int64(N), ct.c_int64(R),
        ct.c_int(M), _d(out), ct.c_int(nthreads),
    )
    return out[0] if squeeze else out


def resample_vals(x, u, freq, order, w=None, nthreads=1):
    """cmomy.wrap_resample_vals(...).transpose(rep, ...): returns (nrep, C, 2, K)."""
    x = np.asarray(x, dtype=np.float64)
    squeeze = x.ndim == 1
    x2 = np.ascontiguousarray(x.reshape(x.shape[0], -1))
    u = _f64(u)
    freq = np.ascontiguousarray(freq, dtype=np.int64)
    N, C = x2.shape
    nrep = freq.shape[0]
    assert freq.shape == (nrep, N)
    K = order + 1
    out = np.zeros((nrep, C, 2, K))
    wp = _d(_f64(w)) if w is not None else None
    ls, lc = _strides(x2)
    lib().orc_resample_vals(
        _d(x2), ct.c_int64(ls), ct.c_int64(lc), _d(u), wp, _i(freq),
        ct.c_int64(N), ct.c_int64(C), ct.c_int64(nrep), ct.c_int(order), _d(out),
        ct.c_int(nthreads),
    )
    return out[:, 0] if squeeze else out


def reduce_data(data, order):
    """CentralMomentsData.reduce(dim=rec): data (nrec, C, 2, K) -> (C, 2, K)."""
    data = _f64(data)
    nrec, C = data.shape[:2]
    out = np.zeros((C, 2, order + 1))
    lib().orc_reduce_data(_d(data), ct.c_int64(nrec), ct.c_int64(C), ct.c_int(order), _d(out))
    return out


def resample_data(data, freq, order):
    """CentralMomentsData.resample_and_reduce: (nrec, C, 2, K), (nrep, nrec) -> (nrep, C, 2, K)."""
    data = _f64(data)
    freq = np.ascontiguousarray(freq, dtype=np.int64)
    nrec, C = data.shape[:2]
    nrep = freq.shape[0]
    out = np.zeros((nrep, C, 2, order + 1))
    lib().orc_resample_data(
        _d(data), _i(freq), ct.c_int64(nrec), ct.c_int64(C), ct.c_int64(nrep),
        ct.c_int(order), _d(out),
    )
    return out


def convert_cov(states, to_central: bool):
    """central<->raw on (..., 2, K) states ([0,0] = weight carried)."""
    s = _f64(states)
    K = s.shape[-1]
    n = s.size // (2 * K)
    out = np.empty_like(s)
    lib().orc_convert_cov(_d(s), _d(out), ct.c_int64(n), ct.c_int(K - 1), ct.c_int(int(to_central)))
    return out


def convert_1d(states, to_central: bool):
    s = _f64(states)
    M = s.shape[-1]
    n = s.size // M
    out = np.empty_like(s)
    lib().orc_convert_1d(_d(s), _d(out), ct.c_int64(n), ct.c_int(M), ct.c_int(int(to_central)))
    return out


def indices_to_freq(indices, ndat=None):
    idx = np.ascontiguousarray(indices, dtype=np.int64)
    nrep, nsamp = idx.shape
    ndat = nsamp if ndat is None else ndat
    freq = np.zeros((nrep, ndat), dtype=np.int64)
    rc = lib().orc_indices_to_freq(_i(idx), ct.c_int64(nrep), ct.c_int64(nsamp), ct.c_int64(ndat), _i(freq))
    if rc != 0:
        raise ValueError("index out of range")
    return freq


def truth_cov(x, u, order, w=None, freq_row=None):
    """Extended-precision two-pass definition: (C, 2, K)."""
    x = np.asarray(x, dtype=np.float64)
    squeeze = x.ndim == 1
    x2 = np.ascontiguousarray(x.reshape(x.shape[0], -1))
    u = _f64(u)
    N, C = x2.shape
    out = np.zeros((C, 2, order + 1))
    wp = _d(_f64(w)) if w is not None else None
    fr = None
    if freq_row is not None:
        fr_a = np.ascontiguousarray(freq_row, dtype=np.int64)
        fr = _i(fr_a)
    lib().orc_truth_cov(
        _d(x2), ct.c_int64(C), ct.c_int64(1), _d(u), wp, fr, ct.c_int64(N), ct.c_int64(C),
        ct.c_int(order), _d(out),
    )
    return out[0] if squeeze else out


def truth_cov_multi(x, u, order, freq_rows, w=None, nthreads=0):
    """truth_cov for R frequency rows at once: (R, C, 2, K); (row, column) pairs on `nthreads` threads (0: all cores).
    The arithmetic of every pair is exactly truth_cov's."""
    import os

    x2 = np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(len(u), -1))
    u = _f64(u)
    fr = np.ascontiguousarray(freq_rows, dtype=np.int64)
    R, N = fr.shape
    C = x2.shape[1]
    out = np.zeros((R, C, 2, order + 1))
    wp = _d(_f64(w)) if w is not None else None
    nt = nthreads or min(os.cpu_count() or 1, R * C)
    lib().orc_truth_cov_multi(_d(x2), ct.c_int64(C), ct.c_int64(1), _d(u), wp, _i(fr), ct.c_int64(R), ct.c_int64(N),
                              ct.c_int64(C), ct.c_int(order), ct.c_int(nt), _d(out))
    return out


def truth_1d(u, mom, w=None, freq_row=None):
    u = _f64(u)
    out = np.zeros(mom + 1)
    wp = _d(_f64(w)) if w is not None else None
    fr = None
    if freq_row is not None:
        fr_a = np.ascontiguousarray(freq_row, dtype=np.int64)
        fr = _i(fr_a)
    lib().orc_truth_1d(_d(u), ct.c_int64(1), wp, fr, ct.c_int64(u.shape[0]), ct.c_int(mom + 1), _d(out))
    return out


# ---------------------------------------------------------------------------
# device-sampler restatement (oracle/philox_oracle.c)
# ---------------------------------------------------------------------------
def philox4x32_10(ctr, key):
    c = np.asarray(ctr, dtype=np.uint32)
    k = np.asarray(key, dtype=np.uint32)
    o = np.zeros(4, dtype=np.uint32)
    lib().orc_philox4x32_10(c.ctypes.data_as(c_u32p), k.ctypes.data_as(c_u32p), o.ctypes.data_as(c_u32p))
    return o


def sampler_ntiles(ndat):
    return int(lib().orc_sampler_ntiles(ndat))


def sampler_tile_counts(seed, nrep, ndat, nsamp=0, rep0=0):
    """Row r = replicate rep0 + r of the stream (txm_sampler_spec.rep0)."""
    nt = sampler_ntiles(ndat)
    counts = np.zeros((nrep, nt), dtype=np.uint32)
    rc = lib().orc_sampler_tile_counts_rep0(
        ct.c_uint64(seed), ct.c_int64(nrep), ct.c_int64(ndat), ct.c_int64(nsamp), ct.c_int64(rep0),
        counts.ctypes.data_as(c_u32p),
    )
    if rc != 0:
        raise ValueError(f"sampler geometry unsupported (rc={rc})")
    return counts


def sampler_freq(seed, nrep, ndat, nsamp=0, counts=None, rep0=0):
    if counts is None:
        counts = sampler_tile_counts(seed, nrep, ndat, nsamp, rep0)
    freq = np.zeros((nrep, ndat), dtype=np.int64)
    rc = lib().orc_sampler_freq_rep0(
        ct.c_uint64(seed), ct.c_int64(nrep), ct.c_int64(ndat), ct.c_int64(rep0),
        counts.ctypes.data_as(c_u32p), _i(freq),
    )
    if rc != 0:
        raise ValueError("sampler failed")
    return freq


# ---------------------------------------------------------------------------
# thermoextrap-side restatement (pure numpy)
# ---------------------------------------------------------------------------
def numpy_sampler_indices(rng: np.random.Generator, nrep: int, ndat: int, nsamp: int | None = None):
    """cmomy.factory_sampler({'nrep': nrep}) index draw (verified):
    ``rng.choice(ndat, size=(nrep, nsamp), replace=True)``."""
    nsamp = ndat if nsamp is None else nsamp
    return rng.choice(ndat, size=(nrep, nsamp), replace=True)


def cmom(states):
    """CentralMomentsData.cmom(): [0,0]->1, [1,0]->0, [0,1]->0."""
    out = np.array(states, dtype=np.float64, copy=True)
    out[..., 0, 0] = 1.0
    out[..., 1, 0] = 0.0
    if out.shape[-1] > 1:
        out[..., 0, 1] = 0.0
    return out


def rmom(states):
    """CentralMomentsData.rmom(): raw <x^a u^b>, [0,0] -> 1."""
    out = convert_cov(states, to_central=False)
    out[..., 0, 0] = 1.0
    return out


def moments_to_comoments(m1d, order):
    """1-D central state [order+2] -> comoment state [2, order+1] with x == u
    (data.py:1182-1191):  out[0, j] = m[j], out[1, j] = m[j+1]; out[1, 0] = <u>."""
    m = np.asarray(m1d, dtype=np.float64)
    K = order + 1
    out = np.empty(m.shape[:-1] + (2, K))
    out[..., 0, :] = m[..., :K]
    out[..., 1, :] = m[..., 1 : K + 1]
    return out


def comoments_to_moments_central(c):
    """Inverse for the *cmom()* form (data.py:899-902): du[0..order+1]."""
    c = np.asarray(c, dtype=np.float64)
    K = c.shape[-1]
    out = np.empty(c.shape[:-2] + (K + 1,))
    out[..., :K] = c[..., 0, :]
    out[..., K] = c[..., 1, K - 1]
    return out


def selectors_central(states, deriv_axis=None):
    """(xave, du, dxdu) as data.py:884-909 defines them from a (..., 2, K) state."""
    c = cmom(states)
    xave = np.asarray(states)[..., 1, 0]
    du = c[..., 0, :]
    dxdu = c[..., 1, :]
    if deriv_axis is not None:
        du = np.take(du, 0, axis=deriv_axis)
    return xave, du, dxdu


def selectors_raw(states, deriv_axis=None):
    """(u, xu) as data.py:854-882 defines them."""
    r = rmom(states)
    u = r[..., 0, :]
    xu = r[..., 1, :]
    if deriv_axis is not None:
        u = np.take(u, 0, axis=deriv_axis)
    return u, xu


def ncpu() -> int:
    return os.cpu_count() or 1
