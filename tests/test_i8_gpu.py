"""The int8-sliced bootstrap kernel (txm_resample_i8.hip) against the FP64 kernel run on
the explicit frequency table of the SAME sampler stream (whose bit-exactness against the
CPU restatement oracle/philox_oracle.c is pinned in test_kernels_gpu.py), and against
the oracle itself at small sizes.  TXM_I8=1 forces the int8 path, TXM_I8=0 the FP64 one.
"""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(txm):
    from thermoextrap_amd import engine

    return engine


def data(N, C, seed, heavy=False):
    g = torch.Generator(device="cuda").manual_seed(seed)
    u = 174.85 + 5.31 * torch.randn(N, generator=g, dtype=torch.float64, device="cuda")
    a = torch.linspace(-1.0, 1.0, C, dtype=torch.float64, device="cuda")
    x = 3.0 + a[None, :] * 0.7 + (0.01 + 0.003 * a[None, :]) * u[:, None] \
        + 0.4 * torch.randn(N, C, generator=g, dtype=torch.float64, device="cuda")
    if heavy:  # outliers: one window sees values 1e4 x the spread
        x[N // 3] += 4.0e3
        u[N // 2] += 5.0e4
    return x, u


def scale(x, u, K):
    sc = torch.empty((x.shape[1], 2, K), dtype=torch.float64, device="cuda")
    for b in range(K):
        sc[:, 0, b] = u.std() ** b
        sc[:, 1, b] = x.std(dim=0) * u.std() ** b
    return sc


def err(a, b, sc):
    return ((a - b).abs() / (b.abs() + sc)).max().item()


@pytest.mark.parametrize("N,C,order,nrep,weighted", [
    (1024, 32, 4, 64, False),       # exactly one full tile
    (1500, 32, 4, 70, False),       # sliding partial last tile, ragged replicate group
    (5000, 5, 4, 3, False),         # few columns, few replicates
    (40000, 32, 4, 130, False),     # several tiles, three replicate groups
    (200000, 32, 4, 64, False),     # four scaling windows in four chunks
    (150000, 9, 1, 64, True),       # order 1
    (30000, 32, 5, 64, False),      # order 5: powers 0-2 and 3-5 in two passes
    (30000, 20, 6, 70, True),       # order 6: 0-3 and 4-6
    (9000, 32, 7, 64, False),       # order 7: 0-3 and 4-7
    (12000, 70, 4, 64, False),      # three column groups (32 + 32 + 6)
    (12000, 33, 2, 70, True),       # second group with a single column
    (40000, 17, 3, 64, True),       # order 3 (u-row shares block 4), weights
    (20000, 32, 2, 100, True),      # order 2 (no shared blocks)
    (70001, 1, 4, 65, False),       # 1-D observable
])
def test_i8_matches_fp64_on_same_stream(eng, monkeypatch, N, C, order, nrep, weighted):
    x, u = data(N, C, 5)
    w = None
    if weighted:
        w = 0.25 + torch.rand(N, dtype=torch.float64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(9))
    K = order + 1
    s = eng.DeviceSampler(20261003 + N, nrep, N)
    monkeypatch.setenv("TXM_I8", "1")
    got = eng.resample_vals(x, u, order, sampler=s, w=w)
    again = eng.resample_vals(x, u, order, sampler=s, w=w)
    assert torch.equal(got, again)                       # deterministic
    monkeypatch.setenv("TXM_I8", "0")
    ref_fused = eng.resample_vals(x, u, order, sampler=s, w=w)
    ref = eng.resample_vals(x, u, order, freq=s.freq(), w=w)
    assert got.shape == (nrep, C, 2, K) and torch.isfinite(got).all()
    sc = scale(x, u, K)[None]
    assert err(ref_fused, ref, sc) < 5e-13
    assert err(got, ref, sc) < 5e-13, err(got, ref, sc)
    # replicate weight = sum of f * w, u-row identical across columns
    assert torch.allclose(got[:, :, 0, 0], ref[:, :, 0, 0], rtol=1e-14, atol=0)
    assert (got[:, :, 0, :] == got[:, :1, 0, :]).all()


def test_i8_vs_oracle_truth(eng, monkeypatch, orc):
    """Small case against the long-double two-pass oracle on the materialised frequencies."""
    N, C, order, nrep = 3000, 6, 4, 5
    x, u = data(N, C, 21)
    s = eng.DeviceSampler(77, nrep, N)
    monkeypatch.setenv("TXM_I8", "1")
    got = eng.resample_vals(x, u, order, sampler=s).cpu().numpy()
    f = s.freq().cpu().numpy()
    xh, uh = x.cpu().numpy(), u.cpu().numpy()
    for r in range(nrep):
        truth = orc.truth_cov(xh, uh, order, w=f[r].astype(np.float64))
        sc = scale(x, u, order + 1).cpu().numpy()
        assert (np.abs(got[r] - truth) / (np.abs(truth) + sc)).max() < 1e-12


def test_i8_outliers_and_pivot(eng, monkeypatch, orc):
    """Error model of the fixed-point slicing: one rint per monomial at 2^-51 of the WINDOW
    maximum.  With a 1e4-sigma outlier in u the other 65535 samples of that window lose their
    (du^4 dx)-sized terms below 2^-51 * max: the result stays within 1e-9 of the long-double
    truth relative to the moment itself (FP64 kernel: 1e-13), far below the 1/sqrt(N)
    bootstrap noise; windows without the outlier are unaffected.  A far pivot only costs
    the usual cancellation."""
    N, C, order, nrep = 150000, 32, 4, 64
    x, u = data(N, C, 33, heavy=True)
    s = eng.DeviceSampler(4242, nrep, N)
    monkeypatch.setenv("TXM_I8", "1")
    got = eng.resample_vals(x, u, order, sampler=s)
    f = s.freq()[:2].cpu().numpy()
    xh, uh = x.cpu().numpy(), u.cpu().numpy()
    sc = scale(x, u, order + 1).cpu().numpy()
    for r in range(2):
        truth = orc.truth_cov(xh, uh, order, w=f[r].astype(np.float64))
        e = np.abs(got[r].cpu().numpy() - truth) / (np.abs(truth) + sc)
        assert e.max() < 1e-9, e.max()
        # everything but the highest power of the outlier variable is at FP64 level
        assert e[:, :, :3].max() < 1e-12, e[:, :, :3].max()
    st = eng.reduce_vals(x, u, order)
    piv = torch.cat([st[0, 0, 1:2] + 2.0 * u.std(), st[:, 1, 0] - 3.0 * x.std(dim=0)]).contiguous()
    got2 = eng.resample_vals(x, u, order, sampler=s, pivot=piv)
    assert err(got2, got, torch.as_tensor(sc, device="cuda")[None]) < 5e-6


def test_i8_nsamp_differs_from_ndat(eng, monkeypatch):
    """Replicates of 3 N / 2 and N / 3 draws (cmomy's nsamp): the count-sum correction of the top digit
    and the exact replicate weight depend on the per-tile draw counts, not on N."""
    N, C, order, nrep = 50000, 24, 4, 64
    x, u = data(N, C, 41)
    sc = scale(x, u, order + 1)[None]
    for nsamp in (N * 3 // 2, N // 3):
        s = eng.DeviceSampler(5, nrep, N, nsamp=nsamp)
        monkeypatch.setenv("TXM_I8", "1")
        got = eng.resample_vals(x, u, order, sampler=s)
        monkeypatch.setenv("TXM_I8", "0")
        ref = eng.resample_vals(x, u, order, freq=s.freq())
        assert (got[:, :, 0, 0] == float(nsamp)).all()
        assert err(got, ref, sc) < 5e-13, err(got, ref, sc)


def test_i8_weights_with_zeros_and_constant_columns(eng, monkeypatch):
    """Zero weights, a constant observable (zero spread -> zero column scale) and an observable equal to u."""
    N, C, order, nrep = 30000, 20, 3, 64
    x, u = data(N, C, 43)
    x[:, 3] = 2.5
    x[:, 7] = u
    w = torch.rand(N, dtype=torch.float64, device="cuda")
    w[::5] = 0.0
    s = eng.DeviceSampler(6, nrep, N)
    monkeypatch.setenv("TXM_I8", "1")
    got = eng.resample_vals(x, u, order, sampler=s, w=w)
    monkeypatch.setenv("TXM_I8", "0")
    ref = eng.resample_vals(x, u, order, freq=s.freq(), w=w)
    sc = scale(x, u, order + 1)[None] + 1e-300
    sc[:, 3, 1, :] = u.std() ** torch.arange(order + 1, dtype=torch.float64, device="cuda") * 1e-12 + 1e-300
    assert torch.isfinite(got).all()
    assert (got[:, 3, 1, 0] == 2.5).all() and (got[:, 3, 1, 1:].abs() < 1e-9).all()
    mask = torch.ones(C, dtype=torch.bool, device="cuda")
    mask[3] = False
    assert err(got[:, mask], ref[:, mask], sc[:, mask]) < 5e-13


def test_dispatch_thresholds(eng, monkeypatch):
    monkeypatch.delenv("TXM_I8", raising=False)
    big = 10_000_000
    assert eng.resample_path(big, 32, 1000, 4) == "int8"
    assert eng.resample_path(big, 32, 64, 4) == "int8"
    assert eng.resample_path(big, 32, 48, 4) == "fp64"        # less than one replicate group
    assert eng.resample_path(big, 32, 300, 2) == "fp64"       # order 2 needs >= 384
    assert eng.resample_path(big, 32, 400, 2) == "int8"
    assert eng.resample_path(big, 8, 1000, 4) == "fp64"       # one 16-column FP64 block is cheaper
    assert eng.resample_path(big, 64, 1000, 4) == "int8"      # two column groups
    assert eng.resample_path(big, 40, 1000, 4) == "fp64"      # 8-column tail group
    assert eng.resample_path(big, 32, 1000, 8) == "fp64"      # order 8: FP64 only
    assert eng.resample_path(100_000, 32, 1000, 4) == "fp64"  # short series
    monkeypatch.setenv("TXM_I8", "1")
    assert eng.resample_path(5000, 3, 2, 1) == "int8"
    assert eng.resample_path(500, 3, 2, 1) == "fp64"          # below one sampler tile: never


def test_non_finite_samples_propagate(eng, monkeypatch):
    """A NaN / inf sample poisons exactly what it poisons in the FP64 kernel: its observable column for
    every replicate (0 * NaN inside the contraction), or everything when it sits in u."""
    N, C, order, nrep = 70000, 20, 4, 64
    x, u = data(N, C, 51)
    s = eng.DeviceSampler(8, nrep, N)
    xb = x.clone()
    xb[12345, 3] = float("nan")
    xb[60000, 7] = float("inf")
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("TXM_I8", mode)
        outs[mode] = eng.resample_vals(xb, u, order, sampler=s)
    monkeypatch.setenv("TXM_I8", "0")
    clean = eng.resample_vals(x, u, order, sampler=s)
    sc = scale(x, u, order + 1)[None]
    for mode, o in outs.items():
        assert torch.isnan(o[:, 3, 1, :]).all() and not torch.isfinite(o[:, 7, 1, :]).any(), mode
        good = [c for c in range(C) if c not in (3, 7)]
        assert torch.isfinite(o[:, good]).all(), mode
        assert err(o[:, good], clean[:, good], sc[:, good]) < 5e-13, mode
    ub = u.clone()
    ub[5] = float("inf")
    monkeypatch.setenv("TXM_I8", "1")
    o = eng.resample_vals(x, ub, order, sampler=s)
    assert not torch.isfinite(o[:, :, :, 1:]).any()
