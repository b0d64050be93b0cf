"""The sixteen-wave cut of the int8 bootstrap kernel (txm_resample_i8w.hip; opt-in, TXM_I8W=1) against the eight-wave
kernel BIT FOR BIT and against the oracle's extended-precision definition.

Both kernels accumulate the same int8 digits exactly in int32 and convert them with the same expression at the flush, so
every moment state must be identical -- the u-row digits riding in the dead byte of the words (quads 0 and 1), the
producer / consumer hand-off of a wave pair through LDS progress words, the in-lane power chain and the split region
layout all sit behind that one comparison.  Reference op: cmomy.wrap_resample_vals (thermoextrap data.py:1803-1810)."""

import os

import numpy as np
import pytest
import torch

from test_i8_gpu import data, truth_err, TOL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(txm):
    from thermoextrap_amd import engine

    return engine


def _run(eng, x, u, order, s, w, wide):
    old = os.environ.get("TXM_I8W")
    os.environ["TXM_I8W"] = "1" if wide else "0"
    try:
        with eng.forced_path("int8"):
            out = eng.resample_vals(x, u, order, sampler=s, w=w).clone()
        torch.cuda.synchronize()
        return out
    finally:
        if old is None:
            os.environ.pop("TXM_I8W", None)
        else:
            os.environ["TXM_I8W"] = old


@pytest.mark.parametrize("N,C,order,nrep,weighted", [
    (1024, 32, 4, 64, False),       # exactly one full tile
    (1500, 32, 4, 70, False),       # sliding partial last tile, ragged replicate group
    (40000, 32, 4, 130, False),     # several tiles, three replicate groups
    (200000, 32, 4, 64, False),     # four scaling windows in four chunks
    (300000, 17, 3, 64, True),      # weighted (du AND w staged), 17 columns: quads past C
    (30000, 32, 0, 64, False),      # order 0: the u-row is the single digit set of power 0
    (30000, 32, 5, 64, False),      # two passes: 3 + 3 powers
    (30000, 20, 6, 70, True),       # 4 + 3, weighted
    (9000, 32, 7, 64, False),       # 4 + 4
    (12000, 70, 4, 64, False),      # three column groups (32 + 32 + 6)
    (2_000_000, 32, 4, 128, False), # 489 windows, the throttle between replicate groups
])
def test_sixteen_waves_equal_eight_waves_bit_for_bit(eng, orc, N, C, order, nrep, weighted):
    x, u = data(N, C, 11 + order)
    w = (0.25 + torch.rand(N, dtype=torch.float64, device="cuda")) if weighted else None
    s = eng.DeviceSampler(5, nrep, N)
    a = _run(eng, x, u, order, s, w, wide=True)
    b = _run(eng, x, u, order, s, w, wide=False)
    assert torch.equal(a, b)
    if N <= 40000:  # and the oracle itself on a few replicates
        freq = s.freq()
        assert truth_err(orc, a, x, u, order, freq, [0, nrep // 2, nrep - 1], w=w) < TOL


def test_weighted_order4_stays_on_eight_waves(eng):
    """du and w do not fit next to five power row sets: that launch is served by the eight-wave kernel either way."""
    x, u = data(30000, 32, 3)
    w = 0.25 + torch.rand(30000, dtype=torch.float64, device="cuda")
    s = eng.DeviceSampler(2, 64, 30000)
    assert torch.equal(_run(eng, x, u, 4, s, w, wide=True), _run(eng, x, u, 4, s, w, wide=False))
