"""The count-table kernel (txm_count_table.hip + txm_resample_i8g.hip; reference op cmomy.wrap_resample_vals as called from
thermoextrap data.py:1803-1810, 1354-1366) against the kernel that draws its counts in place and against the ORACLE.

* BIT FOR BIT the fused kernel: both take exact int32 sums of the same fixed-point words per scaling window and flush them
  with the same expression, so forcing one or the other (`path="int8_table"` / `"int8_fused"`) must not move a bit --
  over orders 0-7, weights, column groups, ragged sizes with a slid last tile, replicate counts that fill one, two (one
  workgroup of a single-row-set pass), three (an odd count: a half-empty last workgroup) and more 128-replicate groups, and
  the second sample matrix.
* The oracle: the long-double definition `orc.truth_cov` on the materialised frequency rows of seeded replicates, for the
  shapes the fused kernel's own tests do not reach through the table path (order 0 on two groups per workgroup, the
  second matrix on its own pass).
"""

import numpy as np
import pytest
import torch

from test_i8_gpu import data, scale, truth_err, TOL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(txm):
    from thermoextrap_amd import engine

    return engine


CASES = [
    # N, C, nrep, order, weighted, second matrix
    (300_000, 32, 200, 0, False, False),      # one row set: two groups in one workgroup
    (300_000, 32, 300, 0, True, False),       # three groups: the last workgroup half empty
    (500_000, 64, 129, 0, False, False),      # two column groups, 129 -> 2 groups
    (1_000_003, 24, 100, 0, True, False),     # one group only, slid last tile, 24 of 32 columns
    (300_000, 32, 640, 0, False, False),      # five groups
    (300_000, 32, 200, 1, False, False),
    (300_000, 32, 200, 2, False, False),      # one pass of three row sets
    (300_000, 32, 200, 3, False, False),      # 2 + 2
    (300_000, 32, 130, 4, True, False),       # 3 + 2, weights
    (300_000, 64, 70, 5, False, False),
    (555_555, 40, 64, 2, False, False),       # a narrow tail group behind a full one
    (2_000_000, 32, 257, 6, True, False),     # 3 + 2 + 2
    (300_000, 32, 200, 7, False, False),      # 3 + 3 + 2
    (300_000, 32, 64, 1, False, True),        # second matrix: powers 0-1 + y in one pass
    (400_001, 32, 100, 4, True, True),        # 3 + (2 + y), weights
    (300_000, 28, 90, 6, False, True),        # 3 + 3 + (1 + y); 28 of 32 columns (whole quads: a 30-column row has no room for the last one)
    (300_000, 32, 33, 0, True, True),         # order 0 + y
    (300_000, 20, 40, 7, True, True),
]


@pytest.mark.parametrize("N,C,nrep,order,weighted,withy", CASES)
def test_table_kernel_equals_fused_kernel_bit_for_bit(eng, N, C, nrep, order, weighted, withy):
    x, u = data(N, C, 7)
    w = (torch.rand(N, dtype=torch.float64, device="cuda") + 0.5) if weighted else None
    y = (x * 0.5 + torch.randn_like(x)) if withy else None
    s = eng.DeviceSampler(11, nrep, N, rep0=5)
    r = {}
    for path in ("int8_fused", "int8_table"):
        out = eng.resample_vals(x, u, order, sampler=s, w=w, y=y, path=path)
        r[path] = out if withy else (out, None)
        assert eng.resample_info()["kernel"] == path
    assert torch.equal(r["int8_table"][0], r["int8_fused"][0])
    if withy:
        # the y row set: bit for bit where the fused kernel carried it too, else (order 4: its own order-0 bootstrap) to rounding
        a, b = r["int8_table"][1], r["int8_fused"][1]
        assert torch.equal(a, b) or (a - b).abs().max().item() <= 1e-14 * b.abs().max().item()


@pytest.mark.parametrize("N,C,nrep,order,weighted", [
    (300_000, 32, 300, 0, True),      # two groups per workgroup, odd group count, weights
    (700_000, 32, 256, 0, False),
    (300_000, 32, 200, 4, False),     # default dispatch takes the table kernel from two groups on at order 4
])
def test_table_kernel_vs_oracle(eng, orc, N, C, nrep, order, weighted):
    x, u = data(N, C, 21)
    w = (torch.rand(N, dtype=torch.float64, device="cuda") + 0.5) if weighted else None
    s = eng.DeviceSampler(3, nrep, N)
    got = eng.resample_vals(x, u, order, sampler=s, w=w, path="int8_table")
    assert eng.resample_info()["kernel"] == "int8_table"
    reps = [0, 127, 128, nrep - 1]
    freq = s.freq()
    e = truth_err(orc, got, x, u, order, freq, reps, w=w)
    assert e < TOL, e


def test_default_dispatch_picks_the_documented_kernel(eng):
    """include/txmom.h's paragraph on the two int8 kernels, seen through resample_info() on real calls."""
    x, u = data(800_000, 32, 5)
    for order, nrep, want in ((0, 256, "int8_table"), (2, 200, "int8_table"), (3, 200, "int8_table"), (3, 128, "int8_fused"), (4, 128, "int8_fused"),
                              (4, 200, "int8_table"), (2, 64, "int8_fused"), (6, 130, "int8_fused"), (6, 256, "int8_table")):
        s = eng.DeviceSampler(1, nrep, x.shape[0])
        eng.resample_vals(x, u, order, sampler=s)
        assert eng.resample_info()["kernel"] == want, (order, nrep)
