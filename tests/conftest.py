"""pytest configuration: `gpu` marker, shared fixtures, oracle access.

The oracle (oracle/) is test infrastructure: it is imported here and in the
test modules only.  Product code (thermoextrap_amd/) never imports it.
"""

from __future__ import annotations

import json
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
GOLDEN = Path(__file__).resolve().parent / "golden"
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle

    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def kat():
    return json.loads((GOLDEN / "kat_notebooks.json").read_text())


@pytest.fixture(scope="session")
def idealgas_data():
    d = np.load(GOLDEN / "idealgas_seed0.npz")
    return d["x"], d["u"]


@pytest.fixture(scope="session")
def legacy():
    d = np.load(GOLDEN / "fixture_legacy.npz")
    return {k: d[k] for k in d.files}


@pytest.fixture(scope="session")
def post_data_rng():
    """Factory: numpy Generator in the state the reference notebooks' global rng
    has right after idealgas.generate_data((100000, 1000), ...) on seed 0."""

    def make():
        rng = np.random.default_rng(0)
        rng.bit_generator.advance(100_000 * 1000)
        return rng

    return make


_NUM = r"[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?"


def kat_case(kat, name):
    """Numbers of a notebook whose stored outputs sit in the golden file as text (`<name>_raw_outputs`):
    Temperature_Extrap_Case2/3/4.ipynb and Customized_Derivatives.ipynb print, in this order, the model's derivatives
    (norm=False), its predictions at betas[:4] (volumes[:4]) and the bootstrap std of those predictions; a later cell prints
    the analytic coefficients and the derivatives refitted on sub-samples ("With N_configs = 100000" is the full data)."""
    import re

    text = "\n".join(v for _, v in sorted(kat[name + "_raw_outputs"].items(), key=lambda kv: int(kv[0])))

    def nums(t):
        return [float(v) for v in re.findall(_NUM, t)]

    def after(label, t=text):
        m = re.search(label + r".*?\[(.*?)\]", t, re.S)
        return nums(m.group(1)) if m else None

    out = {"derivs": after(r"Model parameters \(derivatives\):"), "predict4": after(r"Model predictions:"),
           "true_coefs": after(r"True extrapolation coefficients:"), "derivs_N1e5": after(r"With N_configs = 100000:")}
    m = re.search(r"Bootstrapped uncertainties in predictions:.*?\[(.*?)\]", text, re.S)
    out["boot_std4"] = nums(m.group(1)) if m else None
    if out["boot_std4"] is None:  # Case4 prints the bare DataArray of cell 6
        arrs = re.findall(r"array\(\[(.*?)\]\)", text, re.S)
        out["boot_std4"] = nums(arrs[2]) if len(arrs) > 2 else None
    return out


@pytest.fixture(scope="session")
def idealgas_vol5():
    """Customized_Derivatives.ipynb cell 8: idealgas.generate_data((100000, 1000), beta=1, vol=5) on default_rng(0)
    (tests/golden/make_golden.py); x per configuration, W = -1000 x, dx/dq = x."""
    return np.load(GOLDEN / "idealgas_seed0_vol5.npz")["x"]


def rel_close(a, b, sig):
    """a matches b to `sig` printed significant digits (notebook reprs)."""
    a = np.asarray(a, dtype=float)
    b = np.asarray(b, dtype=float)
    tol = 0.5 * 10.0 ** (-(sig - 1))
    return np.all(np.abs(a - b) <= tol * np.maximum(np.abs(b), 1e-300) * 1.2 + 1e-300)


@pytest.fixture(scope="session")
def txm():
    """The product library, GPU box only."""
    import thermoextrap_amd as txa

    txa.require_gpu()
    return txa
