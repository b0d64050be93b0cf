"""Randomised parity sweep of the int8 bootstrap path (every kernel variant: 8 / 4 / 2 / 1 column quads, one and two passes,
the second-matrix row set, replicate offsets, nsamp != ndat, ragged tiles and replicate groups, weights) against the FP64
kernel on the same sampler draw (GPU box):  python tools/i8_fuzz.py [cases] [seed]"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
from thermoextrap_amd import engine as eng  # noqa: E402
from test_i8_gpu import data, err, scale  # noqa: E402

ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst, worst_y = 0.0, 0.0
for k in range(ncase):
    N = int(rng.choice([1024, 1025, 2047, 3000, 8191, 20000, 65536, 65537, 131072, 300000, 1_000_001, 2_500_000]))
    C = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 12, 16, 17, 31, 32, 33, 40, 48, 64, 65]))
    order = int(rng.integers(0, 8))
    nrep = int(rng.choice([1, 2, 63, 64, 65, 100, 128, 200, 257, 400]))
    if N * nrep > 1.2e8:
        nrep = max(1, int(1.2e8 // N))
    weighted = bool(rng.integers(0, 2))
    with_y = rng.random() < 0.35
    rep0 = 0 if rng.random() < 0.6 else int(rng.integers(1, 1 << 31))
    nsamp = 0 if rng.random() < 0.7 else int(N * rng.choice([0.5, 2.0, 3.3]))
    x, u = data(N, C, int(rng.integers(1 << 30)))
    y = data(N, C, int(rng.integers(1 << 30)))[0] if with_y else None
    w = (0.1 + torch.rand(N, dtype=torch.float64, device="cuda")) if weighted else None
    s = eng.DeviceSampler(int(rng.integers(1 << 40)), nrep, N, nsamp=nsamp, rep0=rep0)
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
    worst = max(worst, e)
    # one rint at 2^-51 of the window maximum per monomial: the highest powers of short series (max / typical
    # ~ 1e4 at order 7, a few hundred draws to average over) sit at 1e-12..1e-11, everything else at 1e-14
    tol = 2e-12 * max(1.0, 4.0 ** (order - 5))
    ey = 0.0
    if with_y:
        ey = float(((gy - ry).abs() / (ry.abs() + y.std())).max())
        worst_y = max(worst_y, ey)
    bad = not (e < tol and ey < 1e-12 and torch.isfinite(got).all())
    print(f"{k:3d} N={N:8d} C={C:2d} order={order} nrep={nrep:3d} w={int(weighted)} y={int(with_y)} rep0={rep0:10d} nsamp={nsamp:8d} "
          f"fp64win={info['windows_fp64']:4d}/{info['windows']:4d}: {e:.2e} {ey:.2e}{'   <-- FAIL' if bad else ''}", flush=True)
    assert not bad
print("worst", worst, "worst y", worst_y)
