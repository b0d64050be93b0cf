"""GPU parity tests: HIP kernels (through the C ABI) vs the CPU oracle.

Tolerances (fp64).  Both the HIP path (pivot-shifted power sums, tree/MFMA
summation) and the oracle (sequential Pebay pushes, cmomy's algorithm) carry
rounding error ~eps*sqrt(N) relative to the natural scale of each moment,
    scale[a,b] = sigma_x^a * sigma_u^b,
so agreement is asserted as  |hip - ref| <= RTOL * (|ref| + scale)  with
RTOL = 1e-12 for N <= 1e5 ("moments <= 1e-12 rel"), and both are
also held to the extended-precision definition (oracle.truth_cov).
Index / frequency bookkeeping is compared bit for bit.
"""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

RTOL = 1e-12


def make_data(rng, N, C, kind="idealgas"):
    """u ~ N(174.85, 5.31^2) (the ideal-gas notebook scale: 3 % relative spread,
    the cancellation regime), x_c = a_c + b_c u + noise."""
    if kind == "idealgas":
        u = rng.normal(174.85, 5.31, N)
        a = rng.normal(0.0, 1.0, C)
        b = rng.normal(1e-3, 5e-4, C)
        x = a[None, :] + b[None, :] * u[:, None] + rng.normal(0, 0.05, (N, C))
    else:  # unit uniforms like the reference's FixtureData
        u = rng.random(N)
        x = rng.random((N, C))
    return x, u


def moment_scale(x, u, order, w=None):
    sx = np.std(x, axis=0)
    su = np.std(u)
    sc = np.empty((x.shape[1], 2, order + 1))
    for b in range(order + 1):
        sc[:, 0, b] = su**b
        sc[:, 1, b] = sx * su**b
    return sc


def assert_states_close(got, ref, scale, rtol=RTOL, what=""):
    got = np.asarray(got)
    err = np.abs(got - ref) / (np.abs(ref) + scale)
    assert np.all(np.isfinite(got)), what
    assert err.max() <= rtol, f"{what}: max scaled err {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)}"


def dev(a, dtype=torch.float64):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).cuda()


@pytest.fixture(scope="module")
def eng(txm):
    from thermoextrap_amd import engine

    return engine


# ---------------------------------------------------------------------------
# reduce_vals  (cmomy.wrap_reduce_vals, data.py:1632-1640)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize(
    "N,C,order",
    [(1, 1, 2), (2, 3, 1), (100, 5, 5), (1000, 1, 2), (4097, 8, 4), (20000, 32, 4), (3000, 33, 6),
     (1500, 64, 3), (700, 130, 2), (513, 600, 1), (50000, 2, 8), (777, 7, 0)],
)
@pytest.mark.parametrize("weighted", [False, True])
def test_reduce_vals_rowmajor(eng, orc, N, C, order, weighted):
    rng = np.random.default_rng(N * 31 + C)
    x, u = make_data(rng, N, C)
    w = rng.random(N) + 0.05 if weighted else None
    got = eng.reduce_vals(dev(x), dev(u), order, w=None if w is None else dev(w)).cpu().numpy()
    ref = orc.reduce_vals(x, u, order, w=w)
    truth = orc.truth_cov(x, u, order, w=w)
    sc = moment_scale(x, u, order)
    if N > 2:
        assert_states_close(got, truth, sc, what="hip vs truth")
        assert_states_close(got, ref, sc, rtol=1e-11, what="hip vs pebay oracle")
    else:
        np.testing.assert_allclose(got, truth, rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize("N,C,order", [(1000, 4, 4), (4099, 17, 3), (64, 2, 5)])
def test_reduce_vals_colmajor_and_pitched(eng, orc, N, C, order):
    rng = np.random.default_rng(5)
    x, u = make_data(rng, N, C)
    ref = orc.truth_cov(x, u, order)
    sc = moment_scale(x, u, order)
    # (val, rec) layout: transposed view of a (C, N) array
    xt = dev(x.T.copy())  # (C, N) contiguous
    got = eng.reduce_vals(xt.t(), dev(u), order).cpu().numpy()
    assert_states_close(got, ref, sc, what="colmajor")
    # row pitch > C (a column slice of a wider array)
    wide = np.zeros((N, C + 3))
    wide[:, :C] = x
    got = eng.reduce_vals(dev(wide)[:, :C], dev(u), order).cpu().numpy()
    assert_states_close(got, ref, sc, what="pitched")
    # odd column count / unaligned base -> scalar-load variant
    got = eng.reduce_vals(dev(wide)[:, 1 : C + 1], dev(u), order).cpu().numpy()
    ref2 = orc.truth_cov(wide[:, 1 : C + 1], u, order)
    assert_states_close(got, ref2, moment_scale(wide[:, 1 : C + 1], u, order) + 1e-300, what="unaligned")


def test_reduce_vals_fixture_matches_legacy_raw_moments(eng, orc, legacy):
    """reference tests/test_data.py:7-38: raw moments of FixtureData vs legacy."""
    x, u, order = legacy["x"], legacy["u"], int(legacy["order"])
    st = eng.reduce_vals(dev(x), dev(u), order)
    raw = eng.convert_cov(st, to_central=False).cpu().numpy()
    np.testing.assert_allclose(raw[0, 0, 1:], legacy["raw_u"][1:], rtol=1e-12)
    np.testing.assert_allclose(raw[:, 1, :].T, legacy["raw_xu"], rtol=1e-12)


def test_reduce_vals_notebook_kat(eng, kat, idealgas_data):
    from conftest import rel_close

    x, u = idealgas_data
    st = eng.reduce_vals(dev(x), dev(u), 2).cpu().numpy()
    assert rel_close(st.ravel(), kat["data_org"]["values"], sig=5)


@pytest.mark.parametrize("R,N,mom", [(1, 1000, 5), (16, 5000, 5), (3, 777, 7), (2, 1, 3)])
def test_reduce_vals_1d(eng, orc, R, N, mom):
    rng = np.random.default_rng(R + N)
    u = rng.normal(3.0, 1.5, (R, N))
    got = eng.reduce_vals_1d(dev(u), mom).cpu().numpy()
    for r in range(R):
        t = orc.truth_1d(u[r], mom)
        sc = np.std(u[r]) ** np.arange(mom + 1) if N > 1 else np.ones(mom + 1)
        err = np.abs(got[r] - t) / (np.abs(t) + sc)
        assert err.max() < RTOL


def test_reduce_errors(eng):
    x = torch.zeros((10, 3), dtype=torch.float64, device="cuda")
    u = torch.zeros(10, dtype=torch.float64, device="cuda")
    from thermoextrap_amd import TxmError

    with pytest.raises(TxmError):
        eng.reduce_vals(x, u, 9)  # order > TXM_MAX_ORDER
    with pytest.raises(ValueError):
        eng.reduce_vals(x, u[:5], 2)
    with pytest.raises(TypeError):
        eng.reduce_vals(x.float(), u, 2)


# ---------------------------------------------------------------------------
# conversion
# ---------------------------------------------------------------------------
def test_convert_matches_oracle(eng, orc):
    rng = np.random.default_rng(9)
    x, u = make_data(rng, 500, 6, kind="unit")
    st = orc.reduce_vals(x, u, 5)
    raw_ref = orc.convert_cov(st, False)
    raw = eng.convert_cov(dev(st), False).cpu().numpy()
    np.testing.assert_allclose(raw, raw_ref, rtol=1e-13)
    back = eng.convert_cov(dev(raw_ref), True).cpu().numpy()
    np.testing.assert_allclose(back, orc.convert_cov(raw_ref, True), rtol=1e-12, atol=1e-15)
    m = orc.reduce_vals_1d(u, 6)
    r1 = eng.convert_1d(dev(m), False).cpu().numpy()
    np.testing.assert_allclose(r1, orc.convert_1d(m, False), rtol=1e-13)


# ---------------------------------------------------------------------------
# sampler bookkeeping: bit exact
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("nrep,nsamp,ndat", [(1, 10, 10), (7, 1000, 1000), (3, 5000, 200), (2, 100, 100000)])
def test_indices_to_freq_bit_exact(eng, orc, nrep, nsamp, ndat):
    rng = np.random.default_rng(1)
    idx = rng.choice(ndat, (nrep, nsamp))
    got = eng.indices_to_freq(dev(idx, torch.int64), ndat).cpu().numpy()
    ref = orc.indices_to_freq(idx, ndat)
    assert got.dtype == np.int64 and np.array_equal(got, ref)


def test_indices_out_of_range_raises(eng):
    from thermoextrap_amd import TxmError

    idx = torch.tensor([[0, 1, 5]], dtype=torch.int64, device="cuda")
    with pytest.raises(TxmError):
        eng.indices_to_freq(idx, 5)


@pytest.mark.parametrize("ndat", [1, 7, 1023, 1024, 1025, 4097, 30000, 1 << 20])
def test_device_sampler_bit_exact_vs_cpu_restatement(eng, orc, ndat):
    nrep = 5 if ndat < (1 << 20) else 2
    seed = 0xC0FFEE1234 + ndat
    s = eng.DeviceSampler(seed, nrep, ndat)
    counts = s.counts.cpu().numpy().view(np.uint32)
    ref_counts = orc.sampler_tile_counts(seed, nrep, ndat)
    assert np.array_equal(counts, ref_counts)
    freq = s.freq().cpu().numpy()
    ref = orc.sampler_freq(seed, nrep, ndat, counts=ref_counts)
    assert np.array_equal(freq, ref)
    assert (freq.sum(axis=1) == ndat).all()


def test_device_sampler_large_geometry_bit_exact(eng, orc):
    """ndat large enough for a 15-level tile tree (heap levels + lane-private subtrees) and a partial last tile."""
    ndat, nrep, seed = 16384 * 1024 + 3 * 1024 + 17, 1, 99
    s = eng.DeviceSampler(seed, nrep, ndat)
    counts = s.counts.cpu().numpy().view(np.uint32)
    ref_counts = orc.sampler_tile_counts(seed, nrep, ndat)
    assert np.array_equal(counts, ref_counts)
    assert counts.sum() == ndat


@pytest.mark.parametrize("ndat,nrep,nsamp", [
    (100_000_000, 2, 0),            # the north-star geometry: 17 levels, a 4096-node heap + 32-leaf lane subtrees
    ((1 << 27) + 1, 1, 0),          # one sample beyond a power of two: 18 levels, the last tile holds ONE sample
    (1 << 24, 1, 0),                # exactly dyadic: every split is 1/2
    (3_000_000, 3, 40_000_000),     # nsamp >> ndat on a 12-level tree
    (50_000_000, 2, 1000),          # nsamp << ndat: almost every node is empty or takes the bit rule
])
def test_device_sampler_tile_counts_bit_exact_at_size(eng, orc, ndat, nrep, nsamp):
    """Tile counts of the binomial-splitting tree vs the CPU restatement, bit for bit, at full size (the oracle
    needs ~1 s per replicate at 1e8: it counts bits, it does not bin draws)."""
    seed = 0xABCDEF12345 + ndat
    s = eng.DeviceSampler(seed, nrep, ndat, nsamp=nsamp) if nsamp else eng.DeviceSampler(seed, nrep, ndat)
    counts = s.counts.cpu().numpy().view(np.uint32)
    ref = orc.sampler_tile_counts(seed, nrep, ndat, nsamp)
    assert counts.shape == ref.shape and np.array_equal(counts, ref)
    assert (counts.sum(axis=1, dtype=np.int64) == (nsamp or ndat)).all()


def test_device_sampler_table_does_not_depend_on_the_launch_shape(eng, orc):
    """A node's variate is a function of (level, index, replicate) alone: calls with >= 1024 replicates walk the tile tree with 256
    lanes a replicate, smaller ones with 1024 (and the subtree depth follows the lanes) -- the rows are the same bits, and the
    oracle's."""
    ndat, seed = 2_500_000, 4242
    big = eng.DeviceSampler(seed, 1100, ndat)
    for a in (0, 517, 1095):
        small = eng.DeviceSampler(seed, 5, ndat, rep0=a)
        assert torch.equal(big.counts[a:a + 5], small.counts), a
    ref = orc.sampler_tile_counts(seed, 3, ndat)
    assert np.array_equal(big.counts[:3].cpu().numpy().view(np.uint32), ref)
    for nd in (1500, 70_000):            # shallow trees: fewer subtree roots than lanes, depth-0 subtrees
        b = eng.DeviceSampler(seed + nd, 1024, nd)
        assert np.array_equal(b.counts[:4].cpu().numpy().view(np.uint32), orc.sampler_tile_counts(seed + nd, 4, nd))
        assert torch.equal(b.counts[1000:1003], eng.DeviceSampler(seed + nd, 3, nd, rep0=1000).counts)


def test_device_sampler_matches_committed_stream_vectors(eng):
    """GPU tables vs tests/golden/sampler_stream_v3.json (no oracle in between)."""
    import json
    from pathlib import Path

    g = json.load(open(Path(__file__).parent / "golden" / "sampler_stream_v3.json"))
    for c in g["cases"]:
        s = eng.DeviceSampler(c["seed"], c["nrep"], c["ndat"], nsamp=c["nsamp"], rep0=c.get("rep0", 0))
        assert s.counts.cpu().numpy().view(np.uint32).tolist() == c["counts"]
        f = s.freq().cpu().numpy()
        assert f[:, :48].tolist() == c["freq_head"]
        w = np.arange(1, c["ndat"] + 1)
        assert [int((f[r] * w).sum()) for r in range(c["nrep"])] == c["freq_checksum"]


def test_device_sampler_nsamp(eng, orc):
    s = eng.DeviceSampler(3, 4, 5000, nsamp=12345)
    f = s.freq().cpu().numpy()
    assert (f.sum(axis=1) == 12345).all()
    assert np.array_equal(f, orc.sampler_freq(3, 4, 5000, nsamp=12345))


# ---------------------------------------------------------------------------
# resample_vals (cmomy.wrap_resample_vals, data.py:1803-1810)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize(
    "N,C,order,nrep",
    [(10, 1, 2, 3), (100, 5, 5, 10), (1000, 8, 4, 20), (3000, 32, 4, 70), (2049, 33, 2, 17),
     (1025, 16, 6, 64), (5000, 3, 8, 5), (1, 2, 2, 2)],
)
@pytest.mark.parametrize("weighted", [False, True])
def test_resample_vals_explicit_freq(eng, orc, N, C, order, nrep, weighted):
    rng = np.random.default_rng(N + C + nrep)
    x, u = make_data(rng, N, C)
    w = rng.random(N) + 0.05 if weighted else None
    idx = rng.choice(N, (nrep, N))
    freq = orc.indices_to_freq(idx, N)
    got = eng.resample_vals(dev(x), dev(u), order, freq=dev(freq, torch.int64), w=None if w is None else dev(w))
    got = got.cpu().numpy()
    assert got.shape == (nrep, C, 2, order + 1)
    truth = np.stack([orc.truth_cov(x, u, order, w=w, freq_row=freq[r]) for r in range(nrep)])
    if N > 2:
        sc = moment_scale(x, u, order)[None]
        assert_states_close(got, truth, sc, what="hip vs truth")
        ref = orc.resample_vals(x, u, freq, order, w=w)
        assert_states_close(got, ref, sc, rtol=1e-11, what="hip vs pebay oracle")
    else:
        np.testing.assert_allclose(got, truth, rtol=1e-13, atol=1e-13)


def test_resample_vals_notebook_kat(eng, orc, kat, idealgas_data, post_data_rng):
    """Data_Organization.ipynb cell 35: the reference's own bootstrap replicates."""
    from conftest import rel_close

    x, u = idealgas_data
    rng = post_data_rng()
    idx = orc.numpy_sampler_indices(rng, 3, len(u))
    freq = eng.indices_to_freq(dev(idx, torch.int64), len(u))
    got = eng.resample_vals(dev(x), dev(u), 2, freq=freq).cpu().numpy()
    assert rel_close(got.ravel(), kat["data_org"]["resample_nrep3"], sig=5)


@pytest.mark.parametrize("N,C,order,nrep", [(5000, 32, 4, 40), (1024, 8, 4, 16), (12345, 5, 3, 33), (300, 1, 2, 100)])
def test_resample_vals_device_sampler(eng, orc, N, C, order, nrep):
    """Scale mode: the fused Philox stage must produce exactly the states of the
    materialised frequency table (which is itself bit-exact vs the CPU stream)."""
    rng = np.random.default_rng(N)
    x, u = make_data(rng, N, C)
    s = eng.DeviceSampler(2024 + N, nrep, N)
    got = eng.resample_vals(dev(x), dev(u), order, sampler=s).cpu().numpy()
    freq = s.freq().cpu().numpy()
    assert np.array_equal(freq, orc.sampler_freq(2024 + N, nrep, N))
    truth = np.stack([orc.truth_cov(x, u, order, freq_row=freq[r]) for r in range(nrep)])
    assert_states_close(got, truth, moment_scale(x, u, order)[None], what="fused sampler")
    # identical to the explicit-freq path on the same table, up to summation order
    got2 = eng.resample_vals(dev(x), dev(u), order, freq=dev(freq, torch.int64)).cpu().numpy()
    assert_states_close(got, got2, moment_scale(x, u, order)[None], rtol=1e-13, what="fused vs explicit")


@pytest.mark.parametrize("C", [1, 2, 3, 4, 5, 7, 8])
@pytest.mark.parametrize("order", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("weighted", [False, True])
def test_resample_vals_narrow_states_packed_powers(eng, orc, C, order, weighted):
    """C <= 8 with the device sampler runs the power-packed contraction (2 or 4 powers of du per
    B-operand column); the states must be those of the oracle on the same frequency table, and equal
    the one-power-per-column kernel (explicit-freq path) up to summation order.  N has a ragged tail."""
    N, nrep = 4099 + 1024 * C, 35
    rng = np.random.default_rng(100 * C + order)
    x, u = make_data(rng, N, C)
    w = rng.uniform(0.2, 3.0, N) if weighted else None
    s = eng.DeviceSampler(77 + C, nrep, N)
    got = eng.resample_vals(dev(x), dev(u), order, sampler=s, w=None if w is None else dev(w)).cpu().numpy()
    freq = s.freq().cpu().numpy()
    scale = moment_scale(x, u, order, w)[None]
    truth = np.stack([orc.truth_cov(x, u, order, w=w, freq_row=freq[r]) for r in range(nrep)])
    assert_states_close(got, truth, scale, what="packed powers vs oracle")
    got2 = eng.resample_vals(dev(x), dev(u), order, freq=dev(freq, torch.int64),
                             w=None if w is None else dev(w)).cpu().numpy()
    assert_states_close(got, got2, scale, rtol=1e-13, what="packed vs one power per column")


def test_resample_user_pivot(eng, orc):
    rng = np.random.default_rng(2)
    x, u = make_data(rng, 2000, 4)
    freq = orc.indices_to_freq(rng.choice(2000, (6, 2000)), 2000)
    piv = np.r_[u.mean(), x.mean(axis=0)]
    a = eng.resample_vals(dev(x), dev(u), 4, freq=dev(freq, torch.int64), pivot=dev(piv)).cpu().numpy()
    b = eng.resample_vals(dev(x), dev(u), 4, freq=dev(freq, torch.int64)).cpu().numpy()
    assert_states_close(a, b, moment_scale(x, u, 4)[None], what="pivot independence")


# ---------------------------------------------------------------------------
# resample_data / reduce_data (data.py:1048-1052, 996)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("nrec,C,order,nrep", [(100, 1, 2, 3), (10, 5, 4, 7), (257, 32, 3, 40), (1, 2, 2, 2)])
def test_resample_data(eng, orc, nrec, C, order, nrep):
    rng = np.random.default_rng(nrec)
    nb = 50
    x, u = make_data(rng, nrec * nb, C)
    blocks = np.stack([orc.truth_cov(x[b * nb : (b + 1) * nb], u[b * nb : (b + 1) * nb], order) for b in range(nrec)])
    freq = orc.indices_to_freq(rng.choice(nrec, (nrep, nrec)), nrec)
    got = eng.resample_data(dev(blocks), dev(freq, torch.int64), order).cpu().numpy()
    ff = np.repeat(freq, nb, axis=1)
    truth = np.stack([orc.truth_cov(x, u, order, freq_row=ff[r]) for r in range(nrep)])
    sc = moment_scale(x, u, order)[None]
    assert_states_close(got, truth, sc, rtol=1e-11, what="vs truth")
    assert_states_close(got, orc.resample_data(blocks, freq, order), sc, rtol=1e-11, what="vs pebay merge")
    red = eng.resample_data(dev(blocks), None, order).cpu().numpy()[0]
    assert_states_close(red, orc.truth_cov(x, u, order), sc[0], rtol=1e-11, what="reduce")


def test_block_bootstrap_notebook_kat(eng, orc, kat, idealgas_data, post_data_rng):
    """Data_Organization.ipynb cells 39-41: DataCentralMoments.from_vals(dim=block)
    then .resample({'nrep': 3}) == resample_and_reduce."""
    from conftest import rel_close

    x, u = idealgas_data
    rng = post_data_rng()
    orc.numpy_sampler_indices(rng, 3, len(u))  # cell 35 consumed these draws first
    xx, uu = x.reshape(100, -1), u.reshape(100, -1)
    # reduce along "block": every row is a contiguous series -> colmajor cov kernel per block
    blocks = torch.stack([eng.reduce_vals(dev(xx[b]), dev(uu[b]), 2) for b in range(100)])[:, None]
    assert rel_close(blocks[:4, 0].cpu().numpy().ravel(), kat["data_org"]["block_values_first6"][:24], sig=5)
    idxb = orc.numpy_sampler_indices(rng, 3, 100)
    fb = eng.indices_to_freq(dev(idxb, torch.int64), 100)
    rep = eng.resample_data(blocks, fb, 2).cpu().numpy()
    assert rel_close(rep.ravel(), kat["data_org"]["block_resample_nrep3"], sig=5)
