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
