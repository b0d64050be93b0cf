"""thermoextrap_amd -- MI355X-native central-(co)moment / bootstrap / derivative
hot path of usnistgov/thermoextrap, behind the reference's own API names.

Layout
  csrc/        hand-written HIP kernels (gfx950) + the C ABI (include/txmom.h)
  _lib.py      ctypes binding; no CPU fallback
  engine.py    device-level calls on torch CUDA tensors
"""

from ._lib import TxmError, load, require_gpu  # noqa: F401

__all__ = ["TxmError", "load", "require_gpu"]
