"""thermoextrap_amd -- MI355X-native central-(co)moment / bootstrap / derivative
hot path of usnistgov/thermoextrap, behind the reference's own API names.

Layout
  csrc/        hand-written HIP kernels (gfx950) + the C ABI (include/txmom.h)
  _lib.py      ctypes binding; no CPU fallback
  engine.py    device-level calls on torch CUDA tensors
  moments.py   mirror of the cmomy calls thermoextrap makes (the reference's boundary)
  data.py      DataCentralMoments, DataCentralMomentsVals, DataValues(Central), ...
  symbolic.py  derivative recursion as exact polynomials -> device tables
  models.py    Derivatives, ExtrapModel, StateCollection
  beta.py      factory_derivatives, factory_extrapmodel
"""

from ._lib import TxmError, load, require_gpu  # noqa: F401

_LAZY = {
    "DataCentralMoments": "data", "DataCentralMomentsVals": "data", "DataValues": "data",
    "DataValuesCentral": "data", "DataCallback": "data", "DataCallbackABC": "data", "DataSelector": "data",
    "factory_data_values": "data", "xrwrap_uv": "data", "xrwrap_xv": "data", "xrwrap_alpha": "data",
    "Derivatives": "models", "ExtrapModel": "models", "StateCollection": "models", "PerturbModel": "models",
    "ExtrapWeightedModel": "models", "InterpModel": "models", "InterpModelPiecewise": "models",
    "DataArray": "xrlite", "Dataset": "xrlite",
}
_MODULES = {"stack", "distributed", "gpr_input", "beta", "data", "models", "moments", "idealgas", "symbolic", "engine", "xrlite", "volume", "volume_idealgas", "lnpi"}


def __getattr__(name):
    import importlib

    if name in _LAZY:
        return getattr(importlib.import_module(f"{__name__}.{_LAZY[name]}"), name)
    if name in _MODULES:
        return importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


__all__ = ["TxmError", "load", "require_gpu", *_LAZY, *_MODULES]
