"""Tool-side switches (NOT read by the library): TXM_I8=0/1 in the environment of a tools/ script sets the process-wide
default path through txm_set_resample_path -- what the library itself read from the environment until round 4."""
import os


def apply():
    from thermoextrap_amd import _lib
    e = os.environ.get("TXM_I8")
    if e and e[0] in "01":
        _lib.check(_lib.load().txm_set_resample_path(int(e[0])), "txm_set_resample_path")
