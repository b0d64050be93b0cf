"""Seeded random-shape sweeps of the int8 bootstrap kernels, in the driver's path (round-5 verdict item 6: these lived in
tools/i8g_fuzz.py / tools/i8_fuzz.py, outside pytest, and found that round's only behavioural surprise).

Reference op: cmomy.wrap_resample_vals as called from thermoextrap data.py:1803-1810, 1354-1366.

* table kernel (txm_count_table.hip + txm_resample_i8g.hip) against the fused kernel (txm_resample_i8t.hip): BIT FOR BIT on
  every full 32-column group -- both take exact int32 sums of the same fixed-point words per scaling window and flush them
  with one expression.  A narrow TAIL group (1..16 columns behind full groups) runs the fused kernel's narrow variant in
  either call, in a pass structure that depends on whether the call's kernels carry y: there the two agree to rounding --
  the documented bound 1e-13 of each comoment's natural scale is asserted, not equality (DESIGN 4.2d "Bits").
* both against the FP64 kernel on the same sampler draw at 1e-12 of scale (order <= 5; the top powers of short series carry the
  single rint of the fixed-point word at a larger multiple: 4^(order - 5), as tests/test_i8_gpu.py states it).
* the whole int8 path (8 / 4 / 2 / 1 column quads, one and two passes, second matrix, replicate offsets, nsamp != ndat,
  ragged tiles and replicate groups, weights) against the FP64 kernel.
"""

import random

import numpy as np
import pytest
import torch

from test_i8_gpu import data, err, scale, truth_err, TOL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(txm):
    from thermoextrap_amd import engine

    return engine


def _table_cases(n, seed):
    rng = random.Random(seed)
    out = []
    for it in range(n):
        N = rng.choice([262144, 300_000, 555_555, 786432, 1_000_003, 1_500_000, 2_097_152 + rng.randrange(1, 1024)])
        C = rng.choice([20, 24, 28, 32, 36, 40, 48, 64])
        pad = rng.choice([0, 0, 4, 8])              # row pitch > C
        nrep = rng.choice([1, 33, 64, 100, 127, 128, 129, 200, 256, 257, 300, 384, 385, 512])
        order = rng.randrange(0, 8)
        out.append((it, N, C, pad, nrep, order, rng.random() < 0.4, rng.random() < 0.3, rng.choice([0, 5, 128, 1000])))
    return out


@pytest.mark.parametrize("it,N,C,pad,nrep,order,weighted,withy,rep0", _table_cases(24, 1))
def test_fuzz_table_kernel_equals_fused_and_fp64(eng, it, N, C, pad, nrep, order, weighted, withy, rep0):
    g = torch.Generator(device="cuda").manual_seed(1000 + it)
    u = 3.0 + 2.0 * torch.randn(N, generator=g, dtype=torch.float64, device="cuda")
    xf = torch.randn(N, C + pad, generator=g, dtype=torch.float64, device="cuda") * 0.7 + 0.05 * u[:, None] + 1.5
    x = xf[:, :C]
    w = (torch.rand(N, generator=g, dtype=torch.float64, device="cuda") + 0.5) if weighted else None
    y = None
    if withy:
        yf = torch.randn(N, C + pad, generator=g, dtype=torch.float64, device="cuda") + 0.3 * xf
        y = yf[:, :C]
    s = eng.DeviceSampler(500 + it, nrep, N, rep0=rep0)
    r = {}
    for path in ("int8_fused", "int8_table"):
        out = eng.resample_vals(x, u, order, sampler=s, w=w, y=y, path=path)
        r[path] = out if withy else (out, None)
        assert eng.resample_info()["kernel"] == path
    sc = scale(x, u, order + 1)[None]
    ntail = C % 32 if 0 < C % 32 <= 16 and withy else 0
    cfull = C - ntail
    assert torch.equal(r["int8_table"][0][:, :cfull], r["int8_fused"][0][:, :cfull])
    if ntail:   # the documented exception: to rounding, bounded
        a, b = r["int8_table"][0][:, cfull:], r["int8_fused"][0][:, cfull:]
        assert err(a, b, sc[:, cfull:]) <= 1e-13
    if withy:
        a, b = r["int8_table"][1], r["int8_fused"][1]
        assert torch.equal(a, b) or (a - b).abs().max().item() <= 1e-14 * b.abs().max().item()
    f = eng.resample_vals(x, u, order, sampler=s, w=w, path="fp64")
    tol = 1e-12 * max(1.0, 4.0 ** (order - 5))
    assert err(r["int8_table"][0], f, sc) < tol and err(r["int8_fused"][0], f, sc) < tol


def _path_cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n):
        N = int(rng.choice([1024, 1025, 2047, 3000, 8191, 20000, 65536, 65537, 131072, 300000, 1_000_001]))
        C = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 12, 16, 17, 31, 32, 33, 40, 48, 64, 65]))
        order = int(rng.integers(0, 8))
        nrep = int(rng.choice([1, 2, 63, 64, 65, 100, 128, 200, 257]))
        if N * nrep > 6e7:
            nrep = max(1, int(6e7 // N))
        weighted = bool(rng.integers(0, 2))
        with_y = bool(rng.random() < 0.35)
        rep0 = 0 if rng.random() < 0.6 else int(rng.integers(1, 1 << 31))
        nsamp = 0 if rng.random() < 0.7 else int(N * rng.choice([0.5, 2.0, 3.3]))
        out.append((k, N, C, order, nrep, weighted, with_y, rep0, nsamp, int(rng.integers(1 << 30)), int(rng.integers(1 << 40))))
    return out


@pytest.mark.parametrize("k,N,C,order,nrep,weighted,with_y,rep0,nsamp,dseed,sseed", _path_cases(24, 0))
def test_fuzz_int8_path_against_fp64_kernel(eng, k, N, C, order, nrep, weighted, with_y, rep0, nsamp, dseed, sseed):
    x, u = data(N, C, dseed)
    y = data(N, C, dseed + 1)[0] if with_y else None
    w = (0.1 + torch.rand(N, dtype=torch.float64, device="cuda")) if weighted else None
    s = eng.DeviceSampler(sseed, nrep, N, nsamp=nsamp, rep0=rep0)
    with eng.forced_path("int8"):
        got = eng.resample_vals(x, u, order, sampler=s, w=w, y=y)
        info = eng.resample_info()
    with eng.forced_path("fp64"):
        ref = eng.resample_vals(x, u, order, sampler=s, w=w, y=y)
    gy = ry = None
    if with_y:
        (got, gy), (ref, ry) = got, ref
    assert info["path"] == "int8", info
    e = err(got, ref, scale(x, u, order + 1)[None])
    tol = 2e-12 * max(1.0, 4.0 ** (order - 5))
    assert e < tol and torch.isfinite(got).all(), (e, tol)
    if with_y:
        assert float(((gy - ry).abs() / (ry.abs() + y.std())).max()) < 1e-12


@pytest.mark.parametrize("order,withy,weighted", [(5, False, False), (5, True, True), (6, False, True), (6, True, False),
                                                  (7, False, False), (7, True, True)])
def test_table_kernel_pass_splits_vs_oracle(eng, orc, order, withy, weighted):
    """Orders 5-7 on the table kernel, with and without a second matrix: every pass split -- 3 + 3, 3 + 3 + 1, 3 + 3 + 2 and
    with y 3 + (3 + y) -> 3 + 2 + (1 + y) ... -- meets the ORACLE (long-double definition on the materialised frequency rows)
    directly, not only through bit-equality with the fused kernel (round-5 verdict, "thin spots")."""
    N, C, nrep = 300_000, 32, 200
    x, u = data(N, C, 31 + order)
    w = (torch.rand(N, dtype=torch.float64, device="cuda") + 0.5) if weighted else None
    y = (0.5 * x + torch.randn_like(x)) if withy else None
    s = eng.DeviceSampler(17, nrep, N)
    out = eng.resample_vals(x, u, order, sampler=s, w=w, y=y, path="int8_table")
    assert eng.resample_info()["kernel"] == "int8_table"
    got, gy = out if withy else (out, None)
    freq = s.freq()
    reps = [0, 127, 128, nrep - 1]
    e = truth_err(orc, got, x, u, order, freq, reps, w=w)
    assert e < TOL * max(1.0, 4.0 ** (order - 5)), e
    if withy:
        for r in reps:
            fr = freq[r].to(torch.float64) * (w if w is not None else 1.0)
            want = (fr @ y) / fr.sum()
            assert ((gy[r] - want).abs() / (want.abs() + y.std())).max().item() < 1e-12
