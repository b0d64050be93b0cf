"""The int8-sliced bootstrap kernel (txm_resample_i8.hip) and its precision guard.

Pinned to the ORACLE: the long-double two-pass definition `orc.truth_cov` evaluated on the
materialised frequencies of the sampler stream (whose bit-exactness against the CPU restatement
oracle/philox_oracle.c is pinned in test_kernels_gpu.py).  The FP64 kernel on the explicit
frequency table of the same stream is used as a second, every-replicate reference.
`engine.forced_path("int8" | "fp64")` forces a kernel; everything marked "default dispatch"
runs with the library's own choice, as a user would.
"""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 1e-12   # |hip - truth| <= TOL * (|truth| + sigma_x^a sigma_u^b), DESIGN.md section 2


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


def truth_err(orc, got, x, u, order, freq, reps, w=None, sc=None):
    """max over `reps` of |got[r] - truth_r| / (|truth_r| + sc), truth = long-double definition."""
    xh, uh = x.cpu().numpy(), u.cpu().numpy()
    wh = None if w is None else w.cpu().numpy()
    sc = scale(x, u, order + 1).cpu().numpy() if sc is None else sc
    worst = 0.0
    for r in reps:
        fr = freq[r].cpu().numpy().astype(np.float64)
        t = orc.truth_cov(xh, uh, order, w=fr if wh is None else fr * wh)
        worst = max(worst, float((np.abs(got[r].cpu().numpy() - t) / (np.abs(t) + sc)).max()))
    return worst


@pytest.mark.parametrize("N,C,order,nrep,weighted", [
    (1024, 32, 4, 64, False),       # exactly one full tile
    (1500, 32, 4, 70, False),       # sliding partial last tile, ragged replicate group
    (5000, 5, 4, 3, False),         # few columns, few replicates
    (40000, 32, 4, 130, False),     # several tiles, three replicate groups
    (200000, 32, 4, 64, False),     # four scaling windows in four chunks
    (150000, 9, 1, 64, True),       # order 1
    (30000, 32, 0, 64, False),      # order 0: means only (the volume callback's <dx/dq> bootstrap)
    (9000, 20, 0, 70, True),        # order 0, weighted
    (30000, 32, 5, 64, False),      # order 5: powers 0-2 and 3-5 in two passes
    (30000, 20, 6, 70, True),       # order 6: 0-3 and 4-6
    (9000, 32, 7, 64, False),       # order 7: 0-3 and 4-7
    (12000, 70, 4, 64, False),      # three column groups (32 + 32 + 6)
    (12000, 33, 2, 70, True),       # second group with a single column
    (40000, 17, 3, 64, True),       # order 3 (u-row shares block 4), weights
    (20000, 32, 2, 100, True),      # order 2 (no shared blocks)
    (70001, 1, 4, 65, False),       # 1-D observable
    (50000, 8, 4, 70, False),       # narrow states: four powers per observable column, two row sets
    (50000, 3, 6, 64, True),        # ... order 6 (powers 0-3 and 4-6), weighted
    (33000, 1, 7, 65, True),        # ... order 7, 1-D observable
    (20000, 8, 1, 64, False),       # ... order 1: one row set, two of four power slots used
    (20000, 6, 3, 130, True),       # ... order 3: exactly one row set
    (40000, 16, 4, 70, False),      # 8 < C <= 16: two powers per observable column, three row sets
    (30000, 12, 5, 64, True),       # ... order 5, weighted
    (30000, 9, 7, 65, False),       # ... order 7: four row sets
    (50000, 4, 3, 100, False),      # narrow states, a last replicate group of 33 .. 63 live replicates (config 5's 100 = 64 + 36)
    (30000, 8, 4, 45, True),
    (60001, 12, 2, 127, False),     # ... 63 live, slid last tile
    (40000, 3, 2, 70, True),        # one-quad states with <= 4 powers: four chunk groups of two waves (round 6) -- weighted,
    (300000, 4, 1, 64, True),       # ... several scaling windows (the groups' accumulators meet in the count tile at every flush),
    (9000, 2, 3, 200, False),       # ... four replicate groups, the last one packed
    (70001, 1, 3, 65, False),       # ... 1-D observable, slid last tile
])
def test_i8_matches_fp64_on_same_stream(eng, orc, N, C, order, nrep, weighted):
    x, u = data(N, C, 5)
    w = None
    if weighted:
        w = 0.25 + torch.rand(N, dtype=torch.float64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(9))
    K = order + 1
    s = eng.DeviceSampler(20261003 + N, nrep, N)
    with eng.forced_path("int8"):
        assert eng.resample_path(N, C, nrep, order) == "int8"
        got = eng.resample_vals(x, u, order, sampler=s, w=w)
        again = eng.resample_vals(x, u, order, sampler=s, w=w)
    assert torch.equal(got, again)                       # deterministic
    freq = s.freq()
    with eng.forced_path("fp64"):
        ref_fused = eng.resample_vals(x, u, order, sampler=s, w=w)
        ref = eng.resample_vals(x, u, order, freq=freq, w=w)
    assert got.shape == (nrep, C, 2, K) and torch.isfinite(got).all()
    sc = scale(x, u, K)[None]
    assert err(ref_fused, ref, sc) < 5e-13
    assert err(got, ref, sc) < 5e-13, err(got, ref, sc)
    # ... and against the oracle itself (first and last replicate)
    assert truth_err(orc, got, x, u, order, freq, (0, nrep - 1), w=w) < TOL
    # replicate weight = sum of f * w, u-row identical across columns
    assert torch.allclose(got[:, :, 0, 0], ref[:, :, 0, 0], rtol=1e-14, atol=0)
    assert (got[:, :, 0, :] == got[:, :1, 0, :]).all()


def _fuzz_cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n):
        N = int(rng.choice([1024, 1025, 2047, 3000, 8191, 20000, 65536, 65537, 131072, 300000]))
        C = int(rng.choice([1, 2, 7, 16, 17, 31, 32, 33, 48, 64, 65, 70]))
        order = 1 + k % 7                                   # every order 1..7, both passes of 5..7
        nrep = int(rng.choice([1, 2, 63, 64, 65, 100, 128, 200]))
        if N * C > 6_000_000:                               # keep the long-double oracle within seconds
            N = max(1024, 6_000_000 // C)
        weighted = bool(k % 2)
        nsamp = 0 if rng.random() < 0.6 else int(N * rng.choice([0.5, 2.0, 3.3]))
        out.append(pytest.param(N, C, order, nrep, weighted, nsamp, int(rng.integers(1 << 30)), id=f"fuzz{k}-N{N}-C{C}-o{order}-r{nrep}-w{int(weighted)}-s{nsamp}"))
    return out


@pytest.mark.parametrize("N,C,order,nrep,weighted,nsamp,seed", _fuzz_cases(28, 0))
def test_i8_fuzz_against_oracle_truth(eng, orc, N, C, order, nrep, weighted, nsamp, seed):
    """Randomised shapes (formerly tools/i8_fuzz.py) against the long-double definition: orders 1-7 (both
    passes), weighted or not, one to three column groups, ragged tiles / replicate groups, nsamp != ndat."""
    x, u = data(N, C, seed)
    w = (0.1 + torch.rand(N, dtype=torch.float64, device="cuda")) if weighted else None
    s = eng.DeviceSampler(seed * 7919 + 1, nrep, N, nsamp=nsamp)
    with eng.forced_path("int8"):
        assert eng.resample_path(N, C, nrep, order) == "int8"
        got = eng.resample_vals(x, u, order, sampler=s, w=w)
    assert torch.isfinite(got).all()
    freq = s.freq()
    reps = sorted({0, nrep // 2, nrep - 1})
    e = truth_err(orc, got, x, u, order, freq, reps, w=w)
    assert e < TOL, e
    if not weighted:
        assert (got[:, :, 0, 0] == float(nsamp or N)).all()


def test_i8_vs_oracle_truth(eng, orc):
    """Small case against the long-double two-pass oracle on the materialised frequencies, every replicate."""
    N, C, order, nrep = 3000, 6, 4, 5
    x, u = data(N, C, 21)
    s = eng.DeviceSampler(77, nrep, N)
    with eng.forced_path("int8"):
        got = eng.resample_vals(x, u, order, sampler=s)
    assert truth_err(orc, got, x, u, order, s.freq(), range(nrep)) < TOL


@pytest.mark.parametrize("C", [32, 8, 12])
def test_guard_sends_outlier_windows_to_fp64(eng, orc, C):
    """(C = 8 and 12: the narrow-state variants with four / two powers per observable column.)
    A 1e4-sigma sample in u (and another in x): the windows holding them are scaled by the outlier, so
    the ordinary samples around it would be rounded at ~their own size -- and the replicates that do NOT
    draw the outlier (37 %) consist of exactly those.  The guard hands these windows to the FP64 kernel:
    the result matches the FP64 kernel and the long-double truth at the usual bound, for replicates with
    and without the outlier, with the scale taken from the clean data."""
    N, order, nrep = 150000, 4, 64
    x, u = data(N, C, 33, heavy=True)
    xc, uc = data(N, C, 33, heavy=False)
    sc_clean = scale(xc, uc, order + 1)
    s = eng.DeviceSampler(4242, nrep, N)
    freq = s.freq()
    with eng.forced_path("int8"):
        got = eng.resample_vals(x, u, order, sampler=s)
        info = eng.resample_info(N, C, nrep, order)
    assert info["path"] == "int8" and 2 <= info["windows_fp64"] <= 4, info   # the two outlier windows (+ neighbours at most)
    with eng.forced_path("fp64"):
        ref = eng.resample_vals(x, u, order, freq=freq)
    assert err(got, ref, sc_clean[None]) < 2e-12, err(got, ref, sc_clean[None])
    fu = freq[:, N // 2].cpu().numpy()
    without, with_ = int(np.flatnonzero(fu == 0)[0]), int(np.flatnonzero(fu > 0)[0])
    assert truth_err(orc, got, x, u, order, freq, (without, with_), sc=sc_clean.cpu().numpy()) < TOL
    # clean data of the same shape: nothing is flagged
    with eng.forced_path("int8"):
        eng.resample_vals(xc, uc, order, sampler=s)
        assert eng.resample_info(N, C, nrep, order)["windows_fp64"] == 0
    # a far pivot only costs the usual cancellation
    st = eng.reduce_vals(xc, uc, order)
    piv = torch.cat([st[0, 0, 1:2] + 2.0 * uc.std(), st[:, 1, 0] - 3.0 * xc.std(dim=0)]).contiguous()
    with eng.forced_path("int8"):
        g1 = eng.resample_vals(xc, uc, order, sampler=s)
        g2 = eng.resample_vals(xc, uc, order, sampler=s, pivot=piv)
    # the pivot-shifted monomials are up to (1 + 2)^4 (1 + 3) ~ 3e2 times the central ones and the window scale grows with
    # them: the model bound is ~3e2 x the int8 path's 1e-13 x the cancellation of the back-shift (another ~3e2) ~ 1e-8
    # (round 2 asserted 5e-6 without having measured; measured on MI355X, printed: see the log line)
    e_far = err(g2, g1, sc_clean[None])
    print(f"far-pivot error (int8 path): {e_far:.3e}")
    assert e_far < 1e-8


@pytest.mark.parametrize("kind", ["student_t3", "lognormal_u", "weights_1e-8", "all_windows_flagged"])
def test_guard_heavy_tails_and_wide_weights(eng, orc, kind):
    """Data whose window maximum is far above its typical size everywhere: heavy-tailed u, weights that
    span eight decades.  Whatever the guard decides per window, the result must meet the usual bound."""
    N, C, order, nrep = 200_000, 20, 4, 64
    g = torch.Generator(device="cuda").manual_seed(61)
    z = torch.randn(N, generator=g, dtype=torch.float64, device="cuda")
    w = None
    if kind == "student_t3":
        chi = (torch.randn(N, 3, generator=g, dtype=torch.float64, device="cuda") ** 2).sum(dim=1) / 3.0
        u = 100.0 + 3.0 * z / chi.sqrt()
    elif kind == "lognormal_u":
        u = torch.exp(1.5 * z)
    elif kind == "weights_1e-8":
        u = 174.85 + 5.31 * z
        w = 10.0 ** (-8.0 * torch.rand(N, generator=g, dtype=torch.float64, device="cuda"))
    else:
        u = 174.85 + 5.31 * z
        u[::1000] += 3.0e3                                   # an outlier in every window
    x = 0.3 * torch.sin((u[:, None] - u.mean()) / u.std() + torch.arange(C, device="cuda")[None, :]) \
        + 0.2 * torch.randn(N, C, generator=g, dtype=torch.float64, device="cuda")
    s = eng.DeviceSampler(97, nrep, N)
    with eng.forced_path("int8"):
        got = eng.resample_vals(x, u, order, sampler=s, w=w)
        info = eng.resample_info(N, C, nrep, order)
    if kind == "all_windows_flagged":
        assert info["windows_fp64"] == info["windows"]
    # robust scale: central 98 % of the data (the plain std of heavy-tailed data is itself an outlier statistic)
    uq = torch.quantile(u[:100000], torch.tensor([0.01, 0.99], dtype=torch.float64, device="cuda"))
    su = (uq[1] - uq[0]) / 4.65
    sx = x.std(dim=0)
    K = order + 1
    sc = np.empty((C, 2, K))
    for b in range(K):
        sc[:, 0, b] = float(su) ** b
        sc[:, 1, b] = sx.cpu().numpy() * float(su) ** b
    e = truth_err(orc, got, x, u, order, s.freq(), (0, 1, nrep - 1), w=w, sc=sc)
    assert e < TOL, (kind, e, info)


def test_i8_nsamp_differs_from_ndat(eng):
    """Replicates of 3 N / 2 and N / 3 draws (cmomy's nsamp): the count-sum correction of the top digit
    and the exact replicate weight depend on the per-tile draw counts, not on N."""
    N, C, order, nrep = 50000, 24, 4, 64
    x, u = data(N, C, 41)
    sc = scale(x, u, order + 1)[None]
    for nsamp in (N * 3 // 2, N // 3):
        s = eng.DeviceSampler(5, nrep, N, nsamp=nsamp)
        with eng.forced_path("int8"):
            got = eng.resample_vals(x, u, order, sampler=s)
        with eng.forced_path("fp64"):
            ref = eng.resample_vals(x, u, order, freq=s.freq())
        assert (got[:, :, 0, 0] == float(nsamp)).all()
        assert err(got, ref, sc) < 5e-13, err(got, ref, sc)


def test_i8_weights_with_zeros_and_constant_columns(eng):
    """Zero weights, a constant observable (zero spread -> zero column scale) and an observable equal to u."""
    N, C, order, nrep = 30000, 20, 3, 64
    x, u = data(N, C, 43)
    x[:, 3] = 2.5
    x[:, 7] = u
    w = torch.rand(N, dtype=torch.float64, device="cuda")
    w[::5] = 0.0
    s = eng.DeviceSampler(6, nrep, N)
    with eng.forced_path("int8"):
        got = eng.resample_vals(x, u, order, sampler=s, w=w)
    with eng.forced_path("fp64"):
        ref = eng.resample_vals(x, u, order, freq=s.freq(), w=w)
    sc = scale(x, u, order + 1)[None] + 1e-300
    sc[:, 3, 1, :] = u.std() ** torch.arange(order + 1, dtype=torch.float64, device="cuda") * 1e-12 + 1e-300
    assert torch.isfinite(got).all()
    assert (got[:, 3, 1, 0] == 2.5).all() and (got[:, 3, 1, 1:].abs() < 1e-9).all()
    mask = torch.ones(C, dtype=torch.bool, device="cuda")
    mask[3] = False
    assert err(got[:, mask], ref[:, mask], sc[:, mask]) < 5e-13


def test_dispatch_thresholds(eng):
    with eng.forced_path(None):
        big = 10_000_000
        assert eng.resample_path(big, 32, 1000, 4) == "int8"
        assert eng.resample_path(big, 32, 64, 4) == "int8"
        assert eng.resample_path(big, 32, 32, 4) == "int8"        # long series: from 32 replicates (round-4 sweep)
        assert eng.resample_path(big, 32, 16, 4) == "fp64"
        assert eng.resample_path(big, 32, 32, 1) == "int8"
        assert eng.resample_path(big, 32, 64, 0) == "fp64"        # order 0: a first call pays from 100
        assert eng.resample_path(big, 32, 100, 0) == "int8"
        assert eng.resample_path(300_000, 32, 64, 4) == "int8"    # short series: order >= 3 from 64, orders 1-2 from 128, order 0 from 384
        assert eng.resample_path(300_000, 32, 100, 2) == "fp64"
        assert eng.resample_path(300_000, 32, 128, 2) == "int8"
        assert eng.resample_path(300_000, 32, 300, 0) == "fp64"
        assert eng.resample_path(300_000, 32, 400, 0) == "int8"
        assert eng.resample_path(big, 8, 1000, 4) == "int8"       # narrow states (round 4: chunk groups, packed fill):
        assert eng.resample_path(big, 8, 8, 4) == "int8"          # ... any replicate count on a long series,
        assert eng.resample_path(big, 8, 1000, 1) == "int8"       # ... from order 1,
        assert eng.resample_path(big, 12, 1000, 1) == "int8"
        assert eng.resample_path(big, 12, 1000, 0) == "fp64"      # ... not order 0 (a single 16-column FP64 block)
        assert eng.resample_path(300_000, 8, 100, 4) == "fp64"    # ... short series from 128 replicates
        assert eng.resample_path(300_000, 8, 128, 4) == "int8"
        assert eng.resample_path(1_000_000, 4, 100, 3) == "int8"  # BASELINE config 5's state
        assert eng.resample_path(big, 64, 1000, 4) == "int8"      # two column groups
        assert eng.resample_path(big, 40, 1000, 4) == "int8"      # 8-column tail group: the narrow-state variant behind a full group
        assert eng.resample_path(big, 40, 1000, 0) == "fp64"      # ... which order 0 does not have
        assert eng.resample_path(big, 32, 1000, 8) == "fp64"      # order 8: FP64 only
        assert eng.resample_path(100_000, 32, 1000, 4) == "fp64"  # short series
    with eng.forced_path("int8"):
        assert eng.resample_path(5000, 3, 2, 1) == "int8"
        assert eng.resample_path(500, 3, 2, 1) == "fp64"          # below one sampler tile: never
    with eng.forced_path("fp64"):
        assert eng.resample_path(10_000_000, 32, 1000, 4) == "fp64"


def test_non_finite_samples_propagate(eng):
    """A NaN / inf sample poisons exactly what it poisons in the FP64 kernel: its observable column for
    every replicate (0 * NaN inside the contraction), or everything when it sits in u."""
    N, C, order, nrep = 70000, 20, 4, 64
    x, u = data(N, C, 51)
    s = eng.DeviceSampler(8, nrep, N)
    xb = x.clone()
    xb[12345, 3] = float("nan")
    xb[60000, 7] = float("inf")
    outs = {}
    for mode in ("int8", "fp64"):
        with eng.forced_path(mode):
            outs[mode] = eng.resample_vals(xb, u, order, sampler=s)
    with eng.forced_path("fp64"):
        clean = eng.resample_vals(x, u, order, sampler=s)
    sc = scale(x, u, order + 1)[None]
    for mode, o in outs.items():
        assert torch.isnan(o[:, 3, 1, :]).all() and not torch.isfinite(o[:, 7, 1, :]).any(), mode
        good = [c for c in range(C) if c not in (3, 7)]
        assert torch.isfinite(o[:, good]).all(), mode
        assert err(o[:, good], clean[:, good], sc[:, good]) < 5e-13, mode
    ub = u.clone()
    ub[5] = float("inf")
    with eng.forced_path("int8"):
        o = eng.resample_vals(x, ub, order, sampler=s)
    assert not torch.isfinite(o[:, :, :, 1:]).any()


# ---------------------------------------------------------------------------
# default dispatch, through the drop-in API, on the derivatives (north_star: 1e-10 rel.)
def _derivs_case(kind, N, C, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    z = torch.randn(N, generator=g, dtype=torch.float64, device="cuda")
    w = None
    if kind == "heavy_tail":
        chi = (torch.randn(N, 4, generator=g, dtype=torch.float64, device="cuda") ** 2).sum(dim=1) / 4.0
        u = 175.0 + 5.0 * z / chi.sqrt()                      # Student t, 4 degrees of freedom
    elif kind == "wide_weights":
        u = 175.0 + 5.0 * z
        w = 10.0 ** (-8.0 * torch.rand(N, generator=g, dtype=torch.float64, device="cuda"))   # 1e-8 .. 1
    else:
        u = 175.0 + 5.0 * z
    ph = torch.arange(C, dtype=torch.float64, device="cuda")[None, :] * 0.37
    x = 1.0 + 0.5 * torch.sin((u[:, None] - 175.0) / 5.0 + ph) + 0.2 * torch.randn(N, C, generator=g, dtype=torch.float64, device="cuda")
    return x, u, w


@pytest.mark.parametrize("kind", ["gaussian", "heavy_tail", "wide_weights"])
def test_default_dispatch_derivs_within_1e10_of_oracle(eng, orc, kind):
    """ExtrapModel.resample(device sampler).derivs() with the library's own kernel choice (int8 path + precision
    guard at this shape) against oracle/derivs_oracle.py on the materialised replicate weights: 1e-10 relative
    (north_star), on ordinary, heavy-tailed and wide-weight data."""
    import thermoextrap_amd as xtrap
    from oracle import derivs_oracle as dor
    from thermoextrap_amd.moments import DeviceDataArray

    N, C, order, nrep = 300_000, 20, 4, 64
    x, u, w = _derivs_case(kind, N, C, 71)
    with eng.forced_path(None):
        assert eng.resample_path(N, C, nrep, order) == "int8"
        data_ = xtrap.DataCentralMomentsVals.from_vals(xv=DeviceDataArray(x, ("rec", "val")), uv=DeviceDataArray(u, ("rec",)),
                                                       order=order, central=True,
                                                       weight=None if w is None else DeviceDataArray(w, ("rec",)))
        xem = xtrap.beta.factory_extrapmodel(5.6, data_)
        boot = xem.resample(sampler={"nrep": nrep, "device": True, "seed": 2026})
        got = np.asarray(boot.derivs(norm=False).values)          # (order + 1, rep, val)
        info = eng.resample_info(N, C, nrep, order)
    assert info["path"] == "int8"
    if kind == "gaussian":
        assert info["windows_fp64"] == 0, info
    else:
        assert info["windows_fp64"] > 0, info
    freq = eng.DeviceSampler(2026, nrep, N).freq().cpu().numpy()
    xh, uh = x.cpu().numpy(), u.cpu().numpy() - 175.0           # exact shift (Sterbenz); derivatives are invariant
    wh = None if w is None else w.cpu().numpy()
    # tolerance on a bound (see test_north_star_step_derivs_vs_oracle_fullsize): |got - ref| <= 1e-12 sum_t |c_t prod atoms|
    # for EVERY entry, atoms = the oracle's extended-precision central comoments; 1e-10 relative where kappa <= 100
    from thermoextrap_amd import symbolic as S

    series = xem.derivatives.series
    worst_b = worst_rel = worst_kappa = worst_old = 0.0
    for r in (0, 17, nrep - 1):
        fr = freq[r].astype(np.float64)
        ref = dor.derivs_x_ave(xh, uh, order, w=fr if wh is None else fr * wh)   # (order + 1, C)
        t = orc.truth_cov(xh, uh, order, w=wh, freq_row=freq[r])                 # (C, 2, K)
        atoms = {"x1": t[:, 1, 0], "du": lambda n: t[0, 0, n], "dxdu": lambda n: t[:, 1, n]}
        res = lambda a: atoms[a[0]] if a[0] == "x1" else atoms[a[0]](a[1])       # noqa: E731
        B = np.stack([np.broadcast_to(S.eval_host(series[k], res, absolute=True), (C,)) for k in range(order + 1)])
        err = np.abs(got[:, r, :] - ref)
        assert np.all(err <= 1e-12 * B), (kind, r, (err / B).max(), info)
        kappa, rel = B / np.abs(ref), err / np.abs(ref)
        ok = kappa <= 100.0
        assert np.all(rel[ok] < 1e-10), (kind, r, rel[ok].max(), info)
        # round 3's criterion as well (round-4 advice: the bound above must add coverage, not replace it): 1e-10 relative on
        # EVERY entry of every order, the denominator floored at the order's median |d_k| over the columns.  It holds on HEAD
        # for all three kinds (worst 3e-15 .. 6e-15, GPU run of round 5) -- the switch to the bound in round 4 was made for
        # the N = 1e8 test, where small derivatives sit far below their order's median, not because this assertion failed.
        floor = np.maximum(np.abs(ref), np.median(np.abs(ref), axis=1, keepdims=True))
        rel_old = err / floor
        assert np.all(rel_old < 1e-10), (kind, r, rel_old.max(), info)
        worst_old = max(worst_old, float(rel_old.max()))
        worst_b, worst_kappa, worst_rel = max(worst_b, (err / B).max()), max(worst_kappa, kappa.max()), max(worst_rel, rel[ok].max())
    print(f"{kind}: max |err| / bound {worst_b:.3e}; max rel (kappa <= 100) {worst_rel:.3e}; worst kappa {worst_kappa:.3e}; "
          f"round-3 criterion (median-floored relative error, limit 1e-10): worst {worst_old:.3e}")


def test_from_resample_vals_matches_resample_and_oracle(eng, orc):
    """DataCentralMoments.from_resample_vals (reference data.py:1285-1392) == DataCentralMomentsVals.resample on a
    shared sampler == the oracle's cmomy restatement / long-double truth; explicit indices and device sampler,
    with weights."""
    import thermoextrap_amd as xtrap
    from thermoextrap_amd import moments as cm
    from thermoextrap_amd.xrlite import DataArray

    rng = np.random.default_rng(5)
    N, C, order, nrep = 4000, 3, 3, 12
    u = rng.normal(174.85, 5.31, N)
    x = 0.2 + 1e-3 * u[:, None] + rng.normal(0, 0.05, (N, C))
    w = rng.uniform(0.5, 1.5, N)
    xv, uv = DataArray(x, dims=("rec", "val")), DataArray(u, dims=("rec",))
    idx = rng.choice(N, (nrep, N))
    freq = orc.indices_to_freq(idx, N)
    for weight in (None, w):
        for spec in ({"indices": idx}, {"nrep": nrep, "device": True, "seed": 31}):
            sampler = cm.factory_sampler(spec, data=xv, dim="rec")
            a = xtrap.DataCentralMoments.from_resample_vals(xv=xv, uv=uv, order=order, sampler=sampler, weight=weight,
                                                            dim="rec", central=True)
            b = xtrap.DataCentralMomentsVals.from_vals(xv=xv, uv=uv, order=order, weight=weight, central=True).resample(sampler=sampler)
            av, bv = np.asarray(a.values.values), np.asarray(b.values.values)
            assert a.values.dims == ("rep", "val", "xmom", "umom") and av.shape == (nrep, C, 2, order + 1)
            np.testing.assert_array_equal(av, bv)               # same kernel, same sampler: same bits
            f = freq if "indices" in spec else sampler.freq
            sc = np.abs(x.std(axis=0))[:, None, None] ** np.array([0, 1])[None, :, None] * u.std() ** np.arange(order + 1)[None, None, :]
            ref = orc.resample_vals(x, u, f, order, w=weight)    # Pebay restatement of cmomy
            assert (np.abs(av - ref) / (np.abs(ref) + sc)).max() < 1e-11
            for r in (0, nrep - 1):
                fr = f[r].astype(np.float64)
                t = orc.truth_cov(x, u, order, w=fr if weight is None else fr * weight)
                assert (np.abs(av[r] - t) / (np.abs(t) + sc)).max() < TOL
            # the derived quantities feed ExtrapModel identically
            da = xtrap.beta.factory_extrapmodel(5.6, a).derivs(norm=False).values
            db = xtrap.beta.factory_extrapmodel(5.6, b).derivs(norm=False).values
            np.testing.assert_array_equal(np.asarray(da), np.asarray(db))
