"""The table-fed kernel for NARROW states (txm_resample_i8gn.hip; reference op cmomy.wrap_resample_vals as called from
thermoextrap data.py:1803-1810, 1354-1366) against the quad-sharing variant of the kernel that draws its counts in place and
against the ORACLE.

* BIT FOR BIT the fused kernel: both take exact int32 sums of the same fixed-point words per scaling window and flush them with the
  same expression into the same partial-sum slots, so forcing one or the other (`path="int8_table"` / `"int8_fused"`) must not
  move a bit -- one, two and four column quads, every order 1..7 (two passes where four quads hold five and more powers), weights,
  ragged sizes with a slid last tile, replicate counts that fill less than one, exactly one and several 128-replicate groups,
  stream offsets.
* the oracle: the long-double definition `orc.truth_cov` on the materialised frequency rows of seeded replicates.
"""

import numpy as np
import pytest
import torch

from test_i8_gpu import data, truth_err, TOL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(txm):
    from thermoextrap_amd import engine

    return engine


CASES = [
    # N, C, nrep, order, weighted
    (300_000, 4, 100, 3, False),      # one quad: BASELINE config 5's state shape
    (300_000, 4, 128, 1, True),
    (500_000, 3, 200, 2, False),      # 3 of 4 columns live... (a 3-double row cannot be DMA'd: falls back, see the test)
    (300_000, 1, 64, 4, False),
    (300_000, 2, 300, 7, True),       # one quad, eight powers: every wave a power, both u-row fragments
    (300_000, 8, 200, 4, False),      # two quads: BASELINE config 2's state shape
    (1_000_003, 8, 129, 4, True),     # slid last tile, 129 -> two groups
    (300_000, 6, 70, 5, False),       # 6 columns in an 8-double row? (pitch 6: the second quad's DMA would leave the row -- falls back)
    (300_000, 8, 256, 7, False),      # two quads, eight powers: two row sets per wave
    (300_000, 8, 33, 1, True),
    (300_000, 12, 200, 3, False),     # four quads, four powers: one pass
    (300_000, 16, 150, 4, True),      # four quads, five powers: two passes (4 + 1)
    (400_001, 16, 64, 7, False),      # four quads, eight powers: 4 + 4
    (300_000, 12, 500, 6, True),
]


@pytest.mark.parametrize("N,C,nrep,order,weighted", CASES)
def test_narrow_table_kernel_equals_fused_kernel_bit_for_bit(eng, N, C, nrep, order, weighted):
    x, u = data(N, C, 9)
    w = (torch.rand(N, dtype=torch.float64, device="cuda") + 0.5) if weighted else None
    s = eng.DeviceSampler(13, nrep, N, rep0=5)
    fused = eng.resample_vals(x, u, order, sampler=s, w=w, path="int8_fused")
    assert eng.resample_info()["kernel"] == "int8_fused"
    table = eng.resample_vals(x, u, order, sampler=s, w=w, path="int8_table")
    k = eng.resample_info()["kernel"]
    dma_ok = C % 4 == 0 or x.stride(0) >= (C + 3) // 4 * 4
    assert k == ("int8_table" if (dma_ok and x.stride(0) % 2 == 0) else "int8_fused"), (k, C, x.stride(0))
    assert torch.equal(table, fused)


@pytest.mark.parametrize("N,C,nrep,order,weighted", [
    (300_000, 4, 100, 3, False),
    (700_000, 8, 200, 4, True),
    (300_000, 16, 130, 5, False),
])
def test_narrow_table_kernel_vs_oracle(eng, orc, N, C, nrep, order, weighted):
    x, u = data(N, C, 23)
    w = (torch.rand(N, dtype=torch.float64, device="cuda") + 0.5) if weighted else None
    s = eng.DeviceSampler(3, nrep, N)
    got = eng.resample_vals(x, u, order, sampler=s, w=w, path="int8_table")
    assert eng.resample_info()["kernel"] == "int8_table"
    e = truth_err(orc, got, x, u, order, s.freq(), [0, min(127, nrep - 1), nrep - 1], w=w)
    assert e < TOL * max(1.0, 4.0 ** (order - 5)), e


def test_narrow_table_rows_equal_offset_call_and_padded_rows(eng):
    """rows [a, b) of a narrow table call equal the (b - a)-replicate call at rep0 = a, bit for bit; a padded row pitch is taken as it is."""
    N, C, order = 300_000, 8, 3
    xf, u = data(N, C + 4, 31)
    x = xf[:, :C]                                  # row pitch 12 doubles
    s = eng.DeviceSampler(7, 300, N, rep0=11)
    whole = eng.resample_vals(x, u, order, sampler=s, path="int8_table")
    assert eng.resample_info()["kernel"] == "int8_table"
    part = eng.resample_vals(x, u, order, sampler=s.rows(128, 300), path="int8_table")
    assert torch.equal(whole[128:300], part)
    assert torch.equal(whole, eng.resample_vals(x.contiguous(), u, order, sampler=s, path="int8_fused"))


def test_rule_takes_the_narrow_table_from_two_replicate_groups_on(eng):
    """Left to the library (path=None) a narrow series of at least 786432 samples rides the table kernel from 65 replicates on
    wherever 128-replicate groups pad no worse than 64s, the fused one below and on shorter series (tools/narrow_table_sweep.py,
    profiles/r06_narrow_table_sweep4.txt; txm_resample.hip narrow_table_pays).  The choice does not move a bit."""
    N, C, order = 1_000_000, 8, 4                 # BASELINE config 2's state shape, a tenth of its length
    x, u = data(N, C, 41)
    s = eng.DeviceSampler(5, 200, N)
    auto = eng.resample_vals(x, u, order, sampler=s)
    assert eng.resample_info()["kernel"] == "int8_table"
    assert torch.equal(auto, eng.resample_vals(x, u, order, sampler=s, path="int8_fused"))
    s1 = eng.DeviceSampler(5, 100, N)             # one group of 128
    one = eng.resample_vals(x, u, 3, sampler=s1)
    assert eng.resample_info()["kernel"] == "int8_table"
    assert torch.equal(one, eng.resample_vals(x, u, 3, sampler=s1, path="int8_fused"))
    eng.resample_vals(x, u, order, sampler=eng.DeviceSampler(5, 64, N))
    assert eng.resample_info()["kernel"] == "int8_fused"
    eng.resample_vals(x, u, order, sampler=eng.DeviceSampler(5, 130, N))      # 130 -> 256 against 192
    assert eng.resample_info()["kernel"] == "int8_fused"
    eng.resample_vals(x[:500_000], u[:500_000], order, sampler=eng.DeviceSampler(5, 200, 500_000))
    assert eng.resample_info()["kernel"] == "int8_fused"
