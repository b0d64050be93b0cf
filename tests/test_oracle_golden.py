"""Pin the CPU oracle against the reference's own numbers (CPU only, no GPU).

* seeded notebook outputs of the reference (tests/golden/kat_notebooks.json):
  reduce_vals layout/values, sampler bookkeeping (Generator.choice on the
  continued global rng), sample- and block-level bootstrap replicates;
* the reference's runnable legacy oracle on FixtureData(100, 5, order=5, seed=0)
  (tests/golden/fixture_legacy.npz): raw moments;
* the mathematical definition in extended precision (oracle truth_cov).
"""

import numpy as np
import pytest

from conftest import rel_close


def test_reduce_vals_matches_notebook(orc, kat, idealgas_data):
    # Data_Organization.ipynb cell 24: data.values for order=2
    x, u = idealgas_data
    st = orc.reduce_vals(x, u, order=2)
    assert st.shape == (2, 3)
    assert rel_close(st.ravel(), kat["data_org"]["values"], sig=5)
    # cells 16-22: u, du, xu, dxdu selectors (data.py:854-909)
    uu, xu = orc.selectors_raw(st)
    xave, du, dxdu = orc.selectors_central(st)
    assert rel_close(uu, kat["data_org"]["u"], sig=5)
    assert rel_close(xu, kat["data_org"]["xu"], sig=5)
    np.testing.assert_allclose(du, kat["data_org"]["du"], atol=5e-5)
    np.testing.assert_allclose(dxdu, kat["data_org"]["dxdu"], atol=5e-5)
    np.testing.assert_allclose(xave, kat["data_org"]["xave"][0], atol=5e-5)


def test_vector_observable_matches_notebook(orc, kat, idealgas_data):
    # cell 47: xv = (x, x^2) -> values[vals=2, 2, 3]
    x, u = idealgas_data
    xv = np.stack([x, x**2], axis=1)
    st = orc.reduce_vals(xv, u, order=2)
    assert rel_close(st.ravel(), kat["data_org"]["vec_values"], sig=5)


def test_sampler_and_bootstrap_match_notebook(orc, kat, idealgas_data, post_data_rng):
    """cells 35, 41, 48, 52: the reference's `resample({"nrep": 3})` calls, in
    notebook order, on the global rng continued from the data draw.  Pins
    sampler == Generator.choice(ndat, (nrep, ndat)) and replicate == weighted
    central comoments with weights freq[r, i]."""
    x, u = idealgas_data
    rng = post_data_rng()
    N = len(u)

    # cell 35: sample-level bootstrap
    idx = orc.numpy_sampler_indices(rng, 3, N)
    freq = orc.indices_to_freq(idx, N)
    assert (freq.sum(axis=1) == N).all()
    rep = orc.resample_vals(x, u, freq, order=2)
    assert rel_close(rep.ravel(), kat["data_org"]["resample_nrep3"], sig=5)

    # cell 39/41: 100 blocks of 1000, block bootstrap (resample_and_reduce)
    xx = x.reshape(100, -1)
    uu = u.reshape(100, -1)
    blocks = np.stack([orc.reduce_vals(xx[b], uu[b], order=2) for b in range(100)])[:, None]  # [100, C=1, 2, 3]
    first = blocks[:4, 0].ravel()
    assert rel_close(first, kat["data_org"]["block_values_first6"][: first.size], sig=5)
    idxb = orc.numpy_sampler_indices(rng, 3, 100)
    fb = orc.indices_to_freq(idxb, 100)
    repb = orc.resample_data(blocks, fb, order=2)
    assert rel_close(repb.ravel(), kat["data_org"]["block_resample_nrep3"], sig=5)

    # cell 48: vector observable, sample-level
    xv = np.stack([x, x**2], axis=1)
    idx = orc.numpy_sampler_indices(rng, 3, N)
    freq = orc.indices_to_freq(idx, N)
    repv = orc.resample_vals(xv, u, freq, order=2)
    assert rel_close(repv.ravel(), kat["data_org"]["vec_resample_nrep3"], sig=5)

    # cell 51/52: vector blocks: reduce + block bootstrap
    xvb = xv.reshape(100, -1, 2)
    blocksv = np.stack([orc.reduce_vals(xvb[b], uu[b], order=2) for b in range(100)])  # [100, 2, 2, 3]
    red = orc.reduce_data(blocksv, order=2)
    assert rel_close(red.ravel(), kat["data_org"]["vec_block_reduce"], sig=5)
    idxb = orc.numpy_sampler_indices(rng, 3, 100)
    fb = orc.indices_to_freq(idxb, 100)
    repvb = orc.resample_data(blocksv, fb, order=2)
    assert rel_close(repvb.ravel(), kat["data_org"]["vec_block_resample_nrep3"], sig=5)


def test_raw_moments_match_legacy(orc, legacy):
    # tests/test_data.py:7-38 of the reference: rdata.u / rdata.xu vs legacy buildAvgFuncs
    x, u, order = legacy["x"], legacy["u"], int(legacy["order"])
    st = orc.reduce_vals(x, u, order)
    uu, xu = orc.selectors_raw(st)
    np.testing.assert_allclose(uu[0], legacy["raw_u"], rtol=1e-12)
    np.testing.assert_allclose(xu.T, legacy["raw_xu"], rtol=1e-12)


@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("order", [1, 4, 6])
def test_pebay_restatement_vs_definition(orc, order, weighted):
    rng = np.random.default_rng(10 + order)
    N, C = 5000, 7
    u = rng.normal(174.85, 5.31, N)
    x = 0.3 + 0.01 * u[:, None] + rng.normal(0, 0.2, (N, C))
    w = rng.random(N) + 0.1 if weighted else None
    a = orc.reduce_vals(x, u, order, w=w)
    t = orc.truth_cov(x, u, order, w=w)
    scale = np.abs(t) + np.std(x) * np.std(u) ** np.arange(order + 1)[None, None, :]
    assert np.max(np.abs(a - t) / scale) < 1e-11


def test_gather_equals_freq(orc):
    # the reference's DataValues.resample gathers rows by sampler.indices
    # (data.py:420-431); cmomy's resample_vals weights by freq.  tests/test_data.py:94-112
    rng = np.random.default_rng(3)
    N, C, order, nrep = 300, 4, 4, 6
    u = rng.random(N)
    x = rng.random((N, C))
    idx = rng.choice(N, (nrep, N))
    freq = orc.indices_to_freq(idx, N)
    a = orc.resample_vals(x, u, freq, order)
    b = np.stack([orc.reduce_vals(x[i], u[i], order) for i in idx])
    np.testing.assert_allclose(a, b, rtol=1e-10, atol=1e-13)


def test_convert_roundtrip_and_definition(orc):
    rng = np.random.default_rng(5)
    N, C, order = 1000, 3, 5
    u = rng.random(N) + 1
    x = rng.random((N, C))
    st = orc.reduce_vals(x, u, order)
    raw = orc.convert_cov(st, to_central=False)
    for c in range(C):
        for a in (0, 1):
            for b in range(order + 1):
                if a == 0 and b == 0:
                    continue
                np.testing.assert_allclose(raw[c, a, b], np.mean(x[:, c] ** a * u**b), rtol=1e-12)
    back = orc.convert_cov(raw, to_central=True)
    np.testing.assert_allclose(back, st, rtol=1e-9, atol=1e-12)


def test_x_is_u_moments_to_comoments(orc):
    # data.py:1182-1191: reduce u to order+1 then moments_to_comoments(mom=(1, order))
    rng = np.random.default_rng(6)
    u = rng.normal(3.0, 1.0, 2000)
    order = 4
    m = orc.reduce_vals_1d(u, order + 1)
    co = orc.moments_to_comoments(m, order)
    direct = orc.reduce_vals(u, u, order)
    np.testing.assert_allclose(co, direct, rtol=1e-10, atol=1e-12)
    # and back (data.py:899-902): du[0..order+1] from the cmom() form
    np.testing.assert_allclose(
        orc.comoments_to_moments_central(orc.cmom(co)), np.r_[1.0, 0.0, m[2:]], rtol=1e-10
    )


def test_philox_known_answers(orc):
    # Random123 kat_vectors for philox4x32-10
    kats = [
        ([0, 0, 0, 0], [0, 0], [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]),
        ([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2, [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]),
        ([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0],
         [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]),
    ]
    for ctr, key, want in kats:
        assert [int(v) for v in orc.philox4x32_10(ctr, key)] == want


@pytest.mark.parametrize("ndat", [1, 7, 1023, 1024, 1025, 4097, 30000])
def test_device_sampler_spec_is_exact_multinomial_shape(orc, ndat):
    nrep = 4
    counts = orc.sampler_tile_counts(11, nrep, ndat)
    assert counts.shape == (nrep, orc.sampler_ntiles(ndat))
    assert (counts.sum(axis=1) == ndat).all()
    freq = orc.sampler_freq(11, nrep, ndat, counts=counts)
    assert (freq >= 0).all() and (freq.sum(axis=1) == ndat).all()
    # tile sums agree with the tile counts
    pad = (-ndat) % 1024
    tiles = np.pad(freq, ((0, 0), (0, pad))).reshape(nrep, -1, 1024).sum(axis=2)
    assert (tiles == counts).all()
    # nsamp != ndat
    f2 = orc.sampler_freq(11, 2, ndat, nsamp=3 * ndat + 1)
    assert (f2.sum(axis=1) == 3 * ndat + 1).all()


def test_device_sampler_statistics(orc):
    """Counts behave like multinomial(N; 1/N): mean 1, var 1-1/N, pair cov -1/N,
    and a chi-square over cells pooled across replicates."""
    ndat, nrep = 5000, 400
    f = orc.sampler_freq(2024, nrep, ndat).astype(float)
    assert abs(f.mean() - 1.0) < 1e-12
    v = f.var(axis=0).mean()
    assert abs(v - (1 - 1 / ndat)) < 0.01
    # column totals over replicates ~ Binomial(nrep*ndat, 1/ndat)
    col = f.sum(axis=0)
    chi2 = ((col - nrep) ** 2 / nrep).sum()
    # dof = ndat - 1; 6 sigma band
    assert abs(chi2 - (ndat - 1)) < 6 * np.sqrt(2 * (ndat - 1))
    # distribution of single counts ~ Poisson(1) (binomial(N,1/N))
    hist = np.bincount(f.astype(int).ravel(), minlength=8)[:8] / f.size
    import math

    pois = np.array([math.exp(-1) / math.factorial(k) for k in range(8)])
    assert np.max(np.abs(hist - pois)) < 2e-3


@pytest.mark.parametrize("ndat", [5 * 1024 + 300, 11 * 1024, 3 * 1024 + 1])
def test_device_sampler_tile_counts_are_multinomial(orc, ndat):
    """Stream v3 draws the tile counts by recursive binomial splitting over a count-balanced tree (BTRS per node,
    the bitwise comparison with the size ratio for small nodes): per-tile mean n*p, variance n*p*(1-p) and pair
    covariance -n*p_i*p_j over many replicates, the partial last tile and the non-dyadic splits included."""
    nrep = 3000
    c = orc.sampler_tile_counts(77, nrep, ndat).astype(float)
    nt = c.shape[1]
    size = np.full(nt, 1024.0)
    size[-1] = ndat - 1024 * (nt - 1)
    p = size / ndat
    mean = c.mean(axis=0)
    sd_mean = np.sqrt(ndat * p * (1 - p) / nrep)
    assert np.all(np.abs(mean - ndat * p) < 5 * sd_mean)
    var = c.var(axis=0, ddof=1)
    assert np.all(np.abs(var / (ndat * p * (1 - p)) - 1) < 6 * np.sqrt(2 / nrep))
    cov = np.cov(c.T)
    want = -ndat * np.outer(p, p)
    off = ~np.eye(nt, dtype=bool)
    # sd of a sample covariance ~ sqrt(var_i var_j / nrep)
    sd_cov = np.sqrt(np.outer(var, var) / nrep)
    assert np.all(np.abs(cov - want)[off] < 6 * sd_cov[off])
    # chi-square of the pooled tile totals
    tot = c.sum(axis=0)
    chi2 = ((tot - nrep * ndat * p) ** 2 / (nrep * ndat * p)).sum()
    assert chi2 < (nt - 1) + 6 * np.sqrt(2 * (nt - 1))


@pytest.mark.parametrize("ndat,nsamp", [(2048, 0), (1024 + 100, 0), (1024 + 5, 0), (2048, 16 * 2048), (1024 + 300, 37)])
def test_node_split_matches_the_binomial_pmf(orc, ndat, nsamp):
    """One tree node, many replicates: the left child's count against the exact Binomial(n, A / (A + B)) pmf
    (chi-square over bins of expected count >= 8).  The cases take BTRS at p = 1/2 (n = 2048 and n = 32768), BTRS at a
    small ratio (the 100-sample partial tile: n p = 100), and the bit-comparison rule (n p = 5; n = 37)."""
    from scipy import stats

    nrep = 120000
    c = orc.sampler_tile_counts(31337, nrep, ndat, nsamp)
    n = nsamp or ndat
    assert c.shape == (nrep, 2) and (c.sum(axis=1) == n).all()
    left = c[:, 0].astype(np.int64)
    p = 1024.0 / ndat
    lo, hi = int(left.min()), int(left.max())
    ks = np.arange(lo, hi + 1)
    expect = nrep * stats.binom.pmf(ks, n, p)
    obs = np.bincount(left - lo, minlength=len(ks)).astype(float)
    # pool the tails so that every bin expects >= 8
    keep = expect >= 8
    i0, i1 = np.argmax(keep), len(keep) - np.argmax(keep[::-1])
    e = np.concatenate([[nrep * stats.binom.cdf(ks[i0] - 1, n, p)], expect[i0:i1], [nrep * stats.binom.sf(ks[i1 - 1], n, p)]])
    o = np.concatenate([[obs[:i0].sum()], obs[i0:i1], [obs[i1:].sum()]])
    e, o = e[e > 0], o[e > 0]
    chi2 = ((o - e) ** 2 / e).sum()
    dof = len(e) - 1
    assert chi2 < dof + 5 * np.sqrt(2 * dof), (chi2, dof)
    assert abs(left.mean() - n * p) < 5 * np.sqrt(n * p * (1 - p) / nrep)


def test_tile_counts_dispersion_at_every_tree_level(orc):
    """ndat = 300 tiles + a partial one, nsamp = 4 ndat: at every level of the count-balanced tree the node totals have
    the binomial mean and variance (BTRS runs with n from 1.2e6 down to ~8000)."""
    ndat, nrep = 300 * 1024 + 77, 600
    nsamp = 4 * ndat
    c = orc.sampler_tile_counts(5150, nrep, ndat, nsamp).astype(float)
    nt = c.shape[1]
    size = np.full(nt, 1024.0)
    size[-1] = ndat - 1024 * (nt - 1)
    k = int(np.ceil(np.log2(nt)))
    for l in range(1, k + 1):
        b = (np.arange((1 << l) + 1) * nt) >> l
        tot = np.add.reduceat(np.concatenate([c, np.zeros((nrep, 1))], axis=1), b[:-1], axis=1)[:, b[:-1] < b[1:]]
        p = np.add.reduceat(np.append(size, 0.0), b[:-1])[b[:-1] < b[1:]] / ndat
        z = (tot.mean(axis=0) - nsamp * p) / np.sqrt(nsamp * p * (1 - p) / nrep)
        assert np.all(np.abs(z) < 5.5), (l, np.abs(z).max())
        ratio = tot.var(axis=0, ddof=1) / (nsamp * p * (1 - p))
        assert np.all(np.abs(ratio - 1) < 6.5 * np.sqrt(2 / nrep)), (l, ratio.min(), ratio.max())


def test_sampler_stream_v3_golden(orc):
    """The stream definition is pinned by committed vectors (tests/golden/sampler_stream_v3.json, generated by
    make_sampler_golden.py): the CPU restatement must reproduce them; the GPU is held to the restatement bit for bit
    in test_kernels_gpu.py."""
    import json
    from pathlib import Path

    g = json.load(open(Path(__file__).parent / "golden" / "sampler_stream_v3.json"))
    assert g["stream_version"] == 3
    for c in g["cases"]:
        r0 = c.get("rep0", 0)
        counts = orc.sampler_tile_counts(c["seed"], c["nrep"], c["ndat"], c["nsamp"], rep0=r0)
        assert counts.tolist() == c["counts"]
        f = orc.sampler_freq(c["seed"], c["nrep"], c["ndat"], c["nsamp"], counts=counts, rep0=r0)
        assert f[:, :48].tolist() == c["freq_head"]
        w = np.arange(1, c["ndat"] + 1)
        assert [int((f[r] * w).sum()) for r in range(c["nrep"])] == c["freq_checksum"]
    # the replicate offset (txm_sampler_spec.rep0): rows a..b of a table == the (b - a, rep0 = a) table
    full_c = orc.sampler_tile_counts(2026, 7, 9001)
    full_f = orc.sampler_freq(2026, 7, 9001, counts=full_c)
    for a, b in ((0, 7), (2, 5), (6, 7)):
        part_c = orc.sampler_tile_counts(2026, b - a, 9001, rep0=a)
        assert np.array_equal(part_c, full_c[a:b])
        assert np.array_equal(orc.sampler_freq(2026, b - a, 9001, counts=part_c, rep0=a), full_f[a:b])
    assert not np.array_equal(orc.sampler_tile_counts(2026, 1, 9001, rep0=1), full_c[0:1])
