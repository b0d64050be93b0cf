"""ctypes binding of libtxmom.so (include/txmom.h).

There is no CPU fallback: if the library is missing, or no gfx950 device is
visible when a compute entry point is used, this raises.  Device memory and
streams are borrowed from torch (plumbing only): arrays cross the ABI as raw
device pointers and sizes.
"""

from __future__ import annotations

import ctypes as ct
from pathlib import Path

from ._build import LIB

c_void_p = ct.c_void_p
c_i64 = ct.c_int64
c_int = ct.c_int
c_size = ct.c_size_t


class SamplerSpec(ct.Structure):
    _fields_ = [("seed", ct.c_uint64), ("nrep", c_i64), ("ndat", c_i64), ("nsamp", c_i64), ("rep0", c_i64)]


class ResampleOpts(ct.Structure):
    """txm_resample_opts (include/txmom.h): per-call kernel choice, device info words, the persistent pre-pass
    block and the optional second sample matrix."""

    _fields_ = [("path", ct.c_int32), ("prep_valid", ct.c_int32), ("prep", c_void_p), ("prep_bytes", c_size),
                ("info", c_void_p), ("y", c_void_p), ("ldy_s", c_i64), ("out_y", c_void_p)]


class StatePtrs(ct.Structure):
    _fields_ = [("x", c_void_p), ("u", c_void_p), ("w", c_void_p)]


class Atom(ct.Structure):
    _fields_ = [("src", ct.c_int32), ("pad", ct.c_int32), ("offset", c_i64), ("s_rep", c_i64), ("s_val", c_i64)]


class PolyTable(ct.Structure):
    _fields_ = [
        ("n_funcs", ct.c_int32), ("n_atoms", ct.c_int32), ("n_terms", ct.c_int32), ("n_factors", ct.c_int32),
        ("log_atom", ct.c_int32), ("pad", ct.c_int32),
        ("atoms", c_void_p), ("func_term0", c_void_p), ("func_flags", c_void_p), ("coef", c_void_p),
        ("term_fac0", c_void_p), ("fac_atom", c_void_p), ("fac_pow", c_void_p),
    ]


# name -> (restype, argtypes); every symbol declared in include/txmom.h
SIGNATURES = {
    "txm_abi_version": (c_int, []),
    "txm_sampler_stream_version": (c_int, []),
    "txm_csrc_sha": (ct.c_char_p, []),
    "txm_last_error": (ct.c_char_p, []),
    "txm_init": (c_int, [c_int]),
    "txm_device_count": (c_int, [ct.POINTER(c_int)]),
    "txm_malloc": (c_int, [ct.POINTER(c_void_p), c_size]),
    "txm_free": (c_int, [c_void_p]),
    "txm_memcpy_h2d": (c_int, [c_void_p, c_void_p, c_size, c_void_p]),
    "txm_memcpy_d2h": (c_int, [c_void_p, c_void_p, c_size, c_void_p]),
    "txm_memset": (c_int, [c_void_p, c_int, c_size, c_void_p]),
    "txm_stream_sync": (c_int, [c_void_p]),
    "txm_reduce_vals_ws_bytes": (c_size, [c_i64, c_i64, c_int]),
    "txm_reduce_vals": (c_int, [c_void_p, c_i64, c_i64, c_void_p, c_void_p, c_i64, c_i64, c_int, c_void_p,
                                c_void_p, c_size, c_void_p]),
    "txm_reduce_vals_pivot": (c_int, [c_void_p, c_i64, c_i64, c_void_p, c_i64, c_i64, c_void_p, c_void_p]),
    "txm_reduce_vals_sums": (c_int, [c_void_p, c_i64, c_i64, c_void_p, c_void_p, c_i64, c_i64, c_int, c_void_p, c_void_p,
                                     c_void_p, c_size, c_void_p]),
    "txm_sums_to_state": (c_int, [c_void_p, c_i64, c_void_p, c_i64, c_int, c_void_p, c_void_p]),
    "txm_push_vals_ws_bytes": (c_size, [c_i64, c_i64, c_int]),
    "txm_push_vals": (c_int, [c_void_p, c_void_p, c_i64, c_i64, c_void_p, c_void_p, c_i64, c_i64, c_int, c_void_p, c_size,
                              c_void_p]),
    "txm_reduce_vals_1d_ws_bytes": (c_size, [c_i64, c_i64, c_int]),
    "txm_reduce_vals_1d": (c_int, [c_void_p, c_i64, c_i64, c_void_p, c_i64, c_i64, c_int, c_void_p, c_void_p,
                                   c_size, c_void_p]),
    "txm_indices_to_freq_ws_bytes": (c_size, []),
    "txm_indices_to_freq": (c_int, [c_void_p, c_i64, c_i64, c_i64, c_void_p, c_void_p, c_size, c_void_p]),
    "txm_sampler_ntiles": (c_i64, [c_i64]),
    "txm_sampler_counts_ws_bytes": (c_size, [ct.POINTER(SamplerSpec)]),
    "txm_sampler_tile_counts": (c_int, [ct.POINTER(SamplerSpec), c_void_p, c_void_p, c_size, c_void_p]),
    "txm_sampler_freq": (c_int, [ct.POINTER(SamplerSpec), c_void_p, c_void_p, c_void_p]),
    "txm_sampler_count_table_bytes": (c_size, [c_i64, c_i64]),
    "txm_sampler_count_table": (c_int, [ct.POINTER(SamplerSpec), c_void_p, c_i64, c_i64, c_void_p, c_void_p]),
    "txm_resample_path": (c_int, [c_i64, c_i64, c_i64, c_int]),
    "txm_resample_kernel": (c_int, [c_i64, c_i64, c_i64, c_int, c_int, c_int, c_int]),
    "txm_resample_operands_aligned": (c_int, [c_void_p, c_i64, c_i64, c_void_p, c_i64]),
    "txm_set_resample_path": (c_int, [c_int]),
    "txm_resample_vals_info": (c_int, [c_void_p, c_i64, c_i64, c_i64, c_int, ct.POINTER(c_i64), c_void_p]),
    "txm_resample_prep_bytes": (c_size, [c_i64, c_i64, c_i64, c_int]),
    "txm_resample_y_ws_bytes": (c_size, [c_i64, c_i64, c_i64]),
    "txm_resample_i8_supported": (c_int, [c_i64, c_i64, c_i64, c_int]),
    "txm_resample_vals_ws_bytes": (c_size, [c_i64, c_i64, c_i64, c_int]),
    "txm_resample_vals_ws_bytes_opts": (c_size, [c_i64, c_i64, c_i64, c_int, c_int, c_int]),
    "txm_resample_vals": (c_int, [c_void_p, c_i64, c_i64, c_void_p, c_void_p, c_i64, c_i64, c_int, c_i64,
                                  c_void_p, ct.POINTER(SamplerSpec), c_void_p, c_void_p, c_void_p,
                                  ct.POINTER(ResampleOpts), c_void_p, c_size, c_void_p]),
    "txm_reduce_vals_batched_ws_bytes": (c_size, [c_i64, c_i64, c_i64, c_int]),
    "txm_reduce_vals_batched": (c_int, [ct.POINTER(StatePtrs), c_i64, c_i64, c_i64, c_i64, c_int, c_void_p, c_void_p,
                                        c_size, c_void_p]),
    "txm_resample_vals_batched_ws_bytes": (c_size, [c_i64, c_i64, c_i64, c_i64, c_int]),
    "txm_resample_vals_batched": (c_int, [ct.POINTER(StatePtrs), c_i64, c_i64, c_i64, c_i64, c_int, c_i64, c_void_p,
                                          ct.POINTER(SamplerSpec), c_void_p, c_void_p, c_void_p, c_size, c_void_p]),
    "txm_resample_batched_prep_bytes": (c_size, [c_i64, c_i64, c_i64, c_i64, c_int]),
    "txm_resample_batched_path": (c_int, [c_i64, c_i64, c_i64, c_i64, c_int]),
    "txm_resample_vals_batched_opts": (c_int, [ct.POINTER(StatePtrs), c_i64, c_i64, c_i64, c_i64, c_int, c_i64, c_void_p,
                                               ct.POINTER(SamplerSpec), c_void_p, c_void_p, ct.POINTER(ResampleOpts), c_void_p,
                                               c_size, c_void_p]),
    "txm_resample_data_ws_bytes": (c_size, [c_i64, c_i64, c_int]),
    "txm_resample_data": (c_int, [c_void_p, c_void_p, c_i64, c_i64, c_i64, c_int, c_void_p, c_void_p, c_size,
                                  c_void_p]),
    "txm_convert_cov": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_int, c_void_p]),
    "txm_convert_1d": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_int, c_void_p]),
    "txm_eval_poly": (c_int, [ct.POINTER(PolyTable), c_void_p, ct.c_int32, c_i64, c_i64, c_void_p, c_void_p]),
    "txm_predict_taylor": (c_int, [c_void_p, ct.c_int32, c_i64, c_void_p, c_i64, ct.c_int32, c_void_p, c_void_p]),
    "txm_cov_over_rep": (c_int, [c_void_p, ct.c_int32, c_i64, c_i64, c_void_p, c_void_p]),
    "txm_perturb_ws_bytes": (c_size, [c_i64, c_i64, ct.c_int32, c_i64]),
    "txm_perturb": (c_int, [c_void_p, c_i64, c_void_p, c_i64, c_i64, ct.POINTER(ct.c_double), ct.c_int32, c_void_p,
                            c_i64, c_void_p, c_void_p, c_size, c_void_p]),
}

ABI_VERSION = 2  # include/txmom.h TXM_ABI_VERSION
SAMPLER_STREAM_VERSION = 3  # include/txmom.h TXM_SAMPLER_STREAM_VERSION (tests/golden/sampler_stream_v3.json)
_lib = None
_gpu_ready = False


class TxmError(RuntimeError):
    pass


def load(path: Path | None = None):
    """Load libtxmom.so and bind every entry point.  No GPU needed for this."""
    global _lib
    if _lib is not None:
        return _lib
    import os

    p = Path(path) if path else Path(os.environ.get("TXM_LIBRARY", LIB))  # TXM_LIBRARY: A/B builds of the kernels
    if not p.exists():
        raise TxmError(
            f"{p} is missing: the HIP extension has not been built. "
            "Run `python -m thermoextrap_amd._build` (hipcc, gfx950). There is no CPU fallback."
        )
    # torch must load its HIP runtime first: loading libtxmom (linked against
    # /opt/rocm's libamdhip64) before torch leaves two runtimes in the process and
    # the second one sees no device.
    import torch  # noqa: F401

    lib = ct.CDLL(str(p))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here == ABI drift: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.txm_abi_version() != ABI_VERSION:
        raise TxmError(f"ABI version mismatch: library {lib.txm_abi_version()} vs binding {ABI_VERSION}")
    if lib.txm_sampler_stream_version() != SAMPLER_STREAM_VERSION:
        raise TxmError(f"sampler stream mismatch: library {lib.txm_sampler_stream_version()} vs binding {SAMPLER_STREAM_VERSION}")
    _lib = lib
    return lib


def last_error() -> str:
    return load().txm_last_error().decode(errors="replace")


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise TxmError(f"{what or 'libtxmom'} failed (status {rc}): {last_error()}")


def require_gpu(device: int | None = None) -> None:
    """Bind the calling thread to a gfx950 device or raise.  No fallback."""
    global _gpu_ready
    lib = load()
    import torch

    if not torch.cuda.is_available():
        raise TxmError("no GPU visible to torch: thermoextrap_amd needs an MI355X (gfx950); there is no CPU path")
    dev = torch.cuda.current_device() if device is None else device
    check(lib.txm_init(dev), "txm_init")
    _gpu_ready = True


def gpu_ready() -> bool:
    return _gpu_ready
