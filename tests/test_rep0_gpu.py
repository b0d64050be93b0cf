"""The replicate offset of the sampler stream (txm_sampler_spec.rep0, ABI 2) and what multi-GPU sharding builds on it:
a rank that owns replicates [a, b) of a bootstrap -- or the states [s0, s1) of a collection whose state s owns replicates
s * nrep ... -- must reproduce its slab of the ONE-GPU result.  Everything here runs in one process on one GPU: the
N-rank result is the concatenation of such slabs (tests/test_distributed_cpu.py drives the collective itself over gloo).

Reference semantics being sharded: independent bootstrap draws per state, a serial loop (models.py:614-641)."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(txm):
    from thermoextrap_amd import engine

    return engine


def _data(N, C, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    u = 174.85 + 5.31 * torch.randn(N, generator=g, dtype=torch.float64, device="cuda")
    x = 0.2 + 1e-3 * u[:, None] + 0.05 * torch.randn(N, C, generator=g, dtype=torch.float64, device="cuda")
    return x, u


@pytest.mark.parametrize("ndat,nsamp", [(5000, 0), (3 * 1024 + 1, 0), (70_000, 123_457), (1 << 20, 0)])
def test_sampler_rows_equal_offset_sampler_bit_for_bit(eng, orc, ndat, nsamp):
    """counts and per-sample frequencies of DeviceSampler(seed, b - a, rep0 = a) are rows a:b of DeviceSampler(seed, n),
    and both are the oracle's stream."""
    seed, n = 20261004, 11
    full = eng.DeviceSampler(seed, n, ndat, nsamp=nsamp)
    cf = full.counts.cpu().numpy()
    ff = full.freq().cpu().numpy() if ndat <= 70_000 else None
    for a, b in ((0, n), (3, 8), (10, 11)):
        part = eng.DeviceSampler(seed, b - a, ndat, nsamp=nsamp, rep0=a)
        assert part.rep0 == a
        assert np.array_equal(part.counts.cpu().numpy(), cf[a:b])
        if ff is not None:
            assert np.array_equal(part.freq().cpu().numpy(), ff[a:b])
    if ndat <= 70_000:
        want = orc.sampler_tile_counts(seed, 5, ndat, nsamp, rep0=3)
        assert np.array_equal(cf[3:8].view(np.uint32), want)
        assert np.array_equal(ff[3:8], orc.sampler_freq(seed, 5, ndat, nsamp, counts=want, rep0=3))
    # far end of the stream's replicate range, and the range check
    hi = eng.DeviceSampler(seed, 2, ndat, nsamp=nsamp, rep0=2**32 - 2)
    assert (hi.counts.cpu().numpy().view(np.uint32).sum(axis=1) == (nsamp or ndat)).all()
    with pytest.raises(Exception):
        eng.DeviceSampler(seed, 3, ndat, nsamp=nsamp, rep0=2**32 - 2)


@pytest.mark.parametrize("path,N,C,order,nrep", [("fp64", 40_000, 5, 3, 24), ("fp64", 300_000, 32, 4, 70),
                                                 ("fp64", 3_000_000, 8, 3, 200), ("fp64", 3_000_000, 40, 2, 130),
                                                 ("int8", 300_000, 32, 4, 200), ("int8", 280_000, 8, 4, 130),
                                                 ("int8", 270_000, 32, 6, 70)])
def test_resample_rows_equal_offset_call(eng, path, N, C, order, nrep):
    """resample_vals(sampler(seed, n, rep0 = k)) == rows k : k + n of the nrep-replicate call.  The sampler rows are the
    same bits (previous test); the moment sums of a replicate are formed per fixed block of samples (int8 path: scaling
    windows; FP64 kernel: sample chunks whose number depends on N alone) and added in block order, whatever the launch
    geometry, so the states agree BIT FOR BIT.  (The N = 3e6 FP64 cases are shapes where a chunking chosen "to fill
    the chip" -- rounds 1 and 2 -- differed between the full call and its slabs.)"""
    x, u = _data(N, C, 5)
    seed = 77
    with eng.forced_path(path):
        full = eng.resample_vals(x, u, order, sampler=eng.DeviceSampler(seed, nrep, N))
        assert eng.resample_info()["path"] == path
        for a, b in ((0, nrep), (nrep // 2, nrep // 2 + 7), (nrep - 64 if nrep > 64 else 0, nrep)):
            part = eng.resample_vals(x, u, order, sampler=eng.DeviceSampler(seed, b - a, N, rep0=a))
            assert torch.equal(part, full[a:b]), (path, a, b, (part - full[a:b]).abs().max().item())
    # ... and a different offset is a different draw
    other = eng.resample_vals(x, u, order, sampler=eng.DeviceSampler(seed, 4, N, rep0=1))
    assert not torch.equal(other, full[0:4])


def _collection(xtrap, S, N, C, order, seed):
    from thermoextrap_amd.moments import DeviceDataArray

    sts = []
    for s in range(S):
        x, u = _data(N, C, seed + s)
        d = xtrap.DataCentralMomentsVals.from_vals(xv=DeviceDataArray(x, ("rec", "val")), uv=DeviceDataArray(u + s, ("rec",)),
                                                   order=order, central=True)
        sts.append(xtrap.beta.factory_extrapmodel(1.0 + 0.5 * s, d))
    return xtrap.models.StateCollection(sts)


@pytest.mark.parametrize("batched", [None, False])
def test_state_shards_equal_slices_of_the_unsharded_collection(txm, eng, batched):
    """What StateCollection.resample(sharded=True) computes on rank r -- the sub-collection states[a:b] resampled with
    state0 = a -- is rows a:b of the unsharded result, bit for bit, on the batched path and on the serial loop; and the
    two paths draw the same replicates (state s owns stream replicates s * nrep ...)."""
    xtrap = txm
    S, N, C, order, nrep = 6, 20_000, 3, 3, 12
    coll = _collection(xtrap, S, N, C, order, 300)
    spec = {"nrep": nrep, "seed": 4242, "device": True}
    whole = coll.resample(spec, batched=batched)
    assert (whole._batch is not None) == (batched is None)
    vals = [torch.as_tensor(st.data.dxduave.device_values) for st in whole.states]
    # the shards of 2 ranks, single states (a rank that holds one state runs the batched kernels too: what
    # _resample_sharded asks for with batched=True), everything
    for a, b in ((0, 3), (3, 6), (2, 3), (0, 1), (0, 6)):
        sub = xtrap.models.StateCollection(coll.states[a:b]).resample(spec, batched=True if batched is None else False, state0=a)
        for i, st in enumerate(sub.states):
            assert torch.equal(st.data.dxduave.device_values, vals[a + i]), (batched, a, b, i)
    # without the offset the shard [3, 6) would repeat the draws of states 0..2: the round-2 defect
    wrong = xtrap.models.StateCollection(coll.states[3:6]).resample(spec, batched=batched)
    assert not torch.equal(wrong.states[0].data.dxduave.device_values, vals[3])
    # states are independent draws: no two states share a frequency row
    f = eng.DeviceSampler(4242, S * nrep, N).freq()
    assert not torch.equal(f[:nrep], f[nrep:2 * nrep])
    # batched and serial agree on the draw (same stream replicates); sums differ by rounding at most
    other = coll.resample(spec, batched=False if batched is None else None)
    for s in range(S):
        o = other.states[s].data.dxduave.device_values
        assert ((o - vals[s]).abs() <= 1e-12 * (vals[s].abs() + 1.0)).all()


def test_many_replicates_in_one_collection(txm, eng):
    """S * nrep beyond 65535 (round-2 advice): the batched path splits the states into groups of launches and still
    equals the per-state stream ranges."""
    xtrap = txm
    S, N, C, order, nrep = 3, 2048, 2, 2, 30_000
    coll = _collection(xtrap, S, N, C, order, 900)
    spec = {"nrep": nrep, "seed": 5, "device": True}
    whole = coll.resample(spec)
    assert whole._batch is not None and whole._batch["big"].shape[:2] == (S, nrep)
    s = 2
    x, u = _data(N, C, 900 + s)
    one = eng.resample_vals(x, u + s, order, sampler=eng.DeviceSampler(5, 50, N, rep0=s * nrep + 29_000))
    got = whole.states[s].data.dxduave.device_values[29_000:29_050]
    assert ((one - got).abs() <= 1e-12 * (got.abs() + 1.0)).all()


def test_prep_block_is_computed_once_per_data_object(txm, eng):
    """The int8 path's pre-pass (window table, guard flags) is kept with the data object: the second bootstrap of the
    same object reuses it (info word 3), gives the same bits as a cold call, and an in-place edit of the samples or
    new_like() invalidates it (reference: per-object cache, data.py:285, 844-942)."""
    from thermoextrap_amd.moments import DeviceDataArray

    xtrap = txm
    N, C, order, nrep = 300_000, 32, 4, 128
    x, u = _data(N, C, 1)
    data = xtrap.DataCentralMomentsVals.from_vals(xv=DeviceDataArray(x, ("rec", "val")), uv=DeviceDataArray(u, ("rec",)),
                                                  order=order, central=True)
    spec = {"nrep": nrep, "seed": 9, "device": True}
    with eng.forced_path("int8"):
        a = data.resample(spec).dxduave.device_values.clone()
        i1 = eng.resample_info()
        b = data.resample(spec).dxduave.device_values.clone()
        i2 = eng.resample_info()
        assert i1["path"] == i2["path"] == "int8" and not i1["prep_reused"] and i2["prep_reused"]
        assert torch.equal(a, b)
        prep = data._cache["resample_prep"]
        assert (prep.hits, prep.misses) == (1, 1)
        # another replicate count: the tables do not depend on it (round 5: replicate slabs share one block) -- reused, and the
        # rows are those of a cold call; another ORDER is another set of tables
        c = data.resample({"nrep": 64, "seed": 9, "device": True}).dxduave.device_values
        assert eng.resample_info()["prep_reused"] and c.shape[0] == 64
        assert torch.equal(c, eng.resample_vals(x, u, order, sampler=eng.DeviceSampler(9, 64, N)))
        eng.resample_vals(x, u, order - 1, sampler=eng.DeviceSampler(9, nrep, N), prep=prep)
        assert not eng.resample_info()["prep_reused"]
        data.resample(spec)   # (the block now holds the order - 1 tables: this call recomputes, the one after it reuses)
        assert not eng.resample_info()["prep_reused"]
        x[1234, 5] += 1.0                                                                       # in-place edit
        d = data.resample(spec).dxduave.device_values
        assert not eng.resample_info()["prep_reused"] and not torch.equal(d, a)
        fresh = data.new_like()
        assert "resample_prep" not in fresh._cache
    # a caller-held block through the engine: cold and warm call, same bits
    prep = eng.ResamplePrep()
    smp = eng.DeviceSampler(3, nrep, N)
    with eng.forced_path("int8"):
        r1 = eng.resample_vals(x, u, order, sampler=smp, prep=prep)
        r2 = eng.resample_vals(x, u, order, sampler=smp, prep=prep)
        assert eng.resample_info()["prep_reused"] and torch.equal(r1, r2)
        r3 = eng.resample_vals(x, u, order, sampler=smp)
        assert torch.equal(r1, r3)


def test_prep_block_key_follows_the_callers_tensors_not_the_temporaries(txm, eng):
    """A non-contiguous x (a transposed view) is copied on every call; the caching allocator hands that temporary the
    same address again with version 0.  The block's key therefore describes the CALLER's tensor and holds a reference
    to it: an in-place edit of the source between two calls recomputes the tables (and the result), an untouched source
    reuses them -- through the engine and through the data-object API with a (val, rec) array."""
    from thermoextrap_amd.moments import DeviceDataArray

    N, C, order, nrep = 300_000, 32, 4, 128
    x, u = _data(N, C, 4)
    xt = x.t().contiguous()            # storage (C, N); xt.t() is the (N, C) view the kernels cannot take as it is
    smp = eng.DeviceSampler(17, nrep, N)
    prep = eng.ResamplePrep()
    with eng.forced_path("int8"):
        cold = eng.resample_vals(x, u, order, sampler=smp)
        r1 = eng.resample_vals(xt.t(), u, order, sampler=smp, prep=prep)
        assert not eng.resample_info()["prep_reused"] and torch.equal(r1, cold)
        r2 = eng.resample_vals(xt.t(), u, order, sampler=smp, prep=prep)
        assert eng.resample_info()["prep_reused"] and torch.equal(r2, cold) and (prep.hits, prep.misses) == (1, 1)
        xt[5, 1234] += 1.0e3           # in-place edit of the SOURCE: far outside the old window scale
        x[1234, 5] += 1.0e3
        r3 = eng.resample_vals(xt.t(), u, order, sampler=smp, prep=prep)
        assert not eng.resample_info()["prep_reused"] and prep.misses == 2
        assert torch.equal(r3, eng.resample_vals(x, u, order, sampler=smp)) and not torch.equal(r3, cold)
        # one block, another array of the same shape: never a hit
        x_other, _ = _data(N, C, 5)
        r4 = eng.resample_vals(x_other, u, order, sampler=smp, prep=prep)
        assert not eng.resample_info()["prep_reused"] and torch.equal(r4, eng.resample_vals(x_other, u, order, sampler=smp))
        # prep.invalidate(): for writes torch does not see
        eng.resample_vals(x_other, u, order, sampler=smp, prep=prep)
        assert eng.resample_info()["prep_reused"]
        prep.invalidate()
        eng.resample_vals(x_other, u, order, sampler=smp, prep=prep)
        assert not eng.resample_info()["prep_reused"]
        # the data-object API on a (val, rec) array (transposed storage)
        data = txm.DataCentralMomentsVals.from_vals(xv=DeviceDataArray(xt, ("val", "rec")), uv=DeviceDataArray(u, ("rec",)),
                                                    order=order, central=True)
        spec = {"nrep": nrep, "seed": 9, "device": True}
        a = data.resample(spec).dxduave.device_values.clone()
        b = data.resample(spec).dxduave.device_values.clone()
        assert eng.resample_info()["prep_reused"] and torch.equal(a, b)
        xt[7, 99] -= 2.0e3
        c = data.resample(spec).dxduave.device_values.clone()
        assert not eng.resample_info()["prep_reused"] and not torch.equal(c, a)


def test_second_matrix_means_equal_separate_order0_bootstrap(eng, orc):
    """txm_resample_opts.y: the per-replicate weighted mean of a second sample matrix on the same draw (the volume
    callback's <dx/dq>, reference volume.py:121-134) equals the mean column of a separate order-0 bootstrap, and the
    oracle's definition on the materialised frequencies."""
    N, C, nrep = 280_000, 32, 128
    x, u = _data(N, C, 8)
    y, _ = _data(N, C, 9)
    g = torch.Generator(device="cuda").manual_seed(4)
    w = 0.25 + torch.rand(N, generator=g, dtype=torch.float64, device="cuda")
    # orders 0..3, 5, 6: the int8 kernel carries y as one more row set of its last pass (one trip over the sampler stream
    # fewer); order 4 (five power row sets) and the FP64 path bootstrap it on their own.  Same numbers either way.
    for order in (4, 2, 6, 0):
        for path in ("int8", "fp64"):
            for ww in (None, w):
                smp = eng.DeviceSampler(12, nrep, N)
                with eng.forced_path(path):
                    st, ym = eng.resample_vals(x, u, order, sampler=smp, w=ww, y=y)
                    st0 = eng.resample_vals(x, u, order, sampler=smp, w=ww)
                    sep = eng.resample_vals(y, u, 0, sampler=smp, w=ww)[:, :, 1, 0]
                assert torch.equal(st, st0), (order, path, ww is not None)
                assert ((ym - sep).abs() <= 1e-13 * (sep.abs() + y.std())).all(), (order, path, ww is not None)
    # wide states (a 32-column group carrying y plus a tail group) and narrow ones (the quad-sharing kernels bootstrap
    # y on their own behind the same entry point)
    for Cw, order in ((40, 2), (40, 5), (8, 3), (12, 2), (3, 4)):
        xw, uw = _data(N, Cw, 18)
        yw_, _ = _data(N, Cw, 19)
        smp = eng.DeviceSampler(13, nrep, N)
        with eng.forced_path("int8"):
            st, ymw = eng.resample_vals(xw, uw, order, sampler=smp, y=yw_)
            assert eng.resample_info()["path"] == "int8"
            st0 = eng.resample_vals(xw, uw, order, sampler=smp)
            sep = eng.resample_vals(yw_, uw, 0, sampler=smp)[:, :, 1, 0]
        if Cw > 32:
            # the full group is bit for bit the call without y; the 8-column tail group runs the wide variant when y rides on
            # the pass and the narrow-state variant (other partial-sum row width, other finalize tree) when it does not
            assert torch.equal(st[:, :32], st0[:, :32]), (Cw, order)
            sc = (xw.std(dim=0)[None, :, None, None] + 1.0) * torch.maximum(st0.abs(), torch.ones_like(st0))
            assert ((st - st0).abs() <= 1e-13 * sc).all(), (Cw, order)
        else:
            assert torch.equal(st, st0), (Cw, order)
        assert ((ymw - sep).abs() <= 1e-13 * (sep.abs() + yw_.std())).all(), (Cw, order)
    order = 4
    f = eng.DeviceSampler(12, 2, N).freq().cpu().numpy()
    yw = (y.cpu().numpy() * (f[1] * w.cpu().numpy())[:, None]).sum(0) / (f[1] * w.cpu().numpy()).sum()
    np.testing.assert_allclose(ym[1].cpu().numpy(), yw, rtol=1e-12)


def test_workspace_budget_slabs_equal_the_unslabbed_rows_bit_for_bit(txm):
    """engine.resample_vals bounds the library workspace (per-window partial sums, the int8 path's count table) by bootstrapping
    the replicates in slabs of whole 128-replicate groups when it exceeds engine.WORKSPACE_BUDGET_BYTES: the slabs are the rows
    of the unslabbed result, bit for bit, on every kernel, with and without a second matrix; one pre-pass block serves them all."""
    import torch

    from thermoextrap_amd import engine as eng

    N, C = 1_200_000, 32
    g = torch.Generator(device="cuda").manual_seed(4)
    u = 174.85 + 5.31 * torch.randn(N, generator=g, dtype=torch.float64, device="cuda")
    x = 0.2 + 1e-3 * u[:, None] + 0.05 * torch.randn(N, C, generator=g, dtype=torch.float64, device="cuda")
    y = 0.5 * x + 1.0
    L = eng._L()
    for nrep, order, path, withy in ((1000, 2, None, False), (1000, 4, None, False), (700, 6, "int8_table", True), (600, 4, "int8_fused", False),
                                    (520, 3, "fp64", False), (900, 1, None, True),
                                    (1000, 0, None, False)):  # order 0: two 128-replicate groups per workgroup, slabs of odd group counts
        s = eng.DeviceSampler(77, nrep, N, rep0=3)
        kw = dict(sampler=s, path=path, y=y if withy else None)
        old = eng.WORKSPACE_BUDGET_BYTES
        try:
            eng.WORKSPACE_BUDGET_BYTES = 1 << 50
            whole = eng.resample_vals(x, u, order, **kw)
            # a budget that forces at least three slabs
            need = L.txm_resample_vals_ws_bytes_opts(N, C, nrep, order, eng._call_path(path), int(withy))
            eng.WORKSPACE_BUDGET_BYTES = need // 3
            assert eng._slab_size(L, N, C, nrep, order, eng._call_path(path), withy) < nrep
            prep = eng.ResamplePrep()
            parts = eng.resample_vals(x, u, order, prep=prep, **kw)
            again = eng.resample_vals(x, u, order, prep=prep, **kw)
        finally:
            eng.WORKSPACE_BUDGET_BYTES = old
        a, b, c = (whole, parts, again) if not withy else (whole[0], parts[0], again[0])
        assert torch.equal(a, b) and torch.equal(a, c), (nrep, order, path)
        if withy:
            assert torch.equal(whole[1], parts[1]), (nrep, order, path)
        if path != "fp64":
            assert prep.misses == 1 and prep.hits >= 3, (prep.misses, prep.hits)


def test_prep_block_is_keyed_on_the_kernel_that_fills_it_second_matrix_order4(txm):
    """Round-5 advice (medium): with y= at order 4 a call of 64 replicates runs the fused kernel, which does NOT carry y (its
    block holds no y tables), a call of 1000 the table kernel, which does.  One ResamplePrep shared by both must not hand the
    second call a block without y's pivot / scales / flags: the key is the kernel word of txm_resample_kernel, so the second
    call recomputes -- and equals a cold call bit for bit, states and y means; coming back to 64 recomputes again."""
    import torch

    from thermoextrap_amd import engine as eng

    N, C, order = 1_000_000, 32, 4
    g = torch.Generator(device="cuda").manual_seed(11)
    u = 174.85 + 5.31 * torch.randn(N, generator=g, dtype=torch.float64, device="cuda")
    x = 0.2 + 1e-3 * u[:, None] + 0.05 * torch.randn(N, C, generator=g, dtype=torch.float64, device="cuda")
    y = 3.0 - 0.25 * x + 0.01 * torch.randn(N, C, generator=g, dtype=torch.float64, device="cuda")
    L = eng._L()
    k64, k1000 = L.txm_resample_kernel(N, C, 64, order, -1, 1, 1), L.txm_resample_kernel(N, C, 1000, order, -1, 1, 1)
    assert k64 == 2 and k1000 == (3 | 0x100)
    prep = eng.ResamplePrep()
    s64, s1000 = eng.DeviceSampler(5, 64, N), eng.DeviceSampler(5, 1000, N)
    a64 = eng.resample_vals(x, u, order, sampler=s64, y=y, prep=prep)
    assert (prep.hits, prep.misses) == (0, 1)
    a1000 = eng.resample_vals(x, u, order, sampler=s1000, y=y, prep=prep)
    assert (prep.hits, prep.misses) == (0, 2) and not eng.resample_info()["prep_reused"]
    b1000 = eng.resample_vals(x, u, order, sampler=s1000, y=y, prep=prep)
    assert (prep.hits, prep.misses) == (1, 2) and eng.resample_info()["prep_reused"]
    c1000 = eng.resample_vals(x, u, order, sampler=s1000, y=y)           # cold: no block at all
    for got in (a1000, b1000):
        assert torch.equal(got[0], c1000[0]) and torch.equal(got[1], c1000[1])
    c64 = eng.resample_vals(x, u, order, sampler=s64, y=y)
    b64 = eng.resample_vals(x, u, order, sampler=s64, y=y, prep=prep)
    assert prep.misses == 3
    assert torch.equal(a64[0], c64[0]) and torch.equal(a64[1], c64[1]) and torch.equal(b64[0], c64[0]) and torch.equal(b64[1], c64[1])
    # the y means are the weighted means of y on the draw, whatever kernel carried them
    f = s1000.freq()[:4].to(torch.float64)
    np.testing.assert_allclose((f @ y / f.sum(1, keepdim=True)).cpu().numpy(), c1000[1][:4].cpu().numpy(), rtol=1e-12)


def test_short_last_slab_takes_the_whole_calls_kernel(txm):
    """Round-5 advice (low): replicate slabs used to pin only int8 vs FP64; a last slab of <= 64 replicates then ran the fused
    kernel while the others ran the table kernel -- with y= at order 4 its y means came from the separate order-0 bootstrap and
    agreed with the unslabbed rows only to rounding.  Slabs now carry the whole call's kernel: 128 + 128 + 128 + 40 replicates,
    y at order 4, bit for bit the unslabbed call, ONE pre-pass block for all of them and for the unslabbed call."""
    import torch

    from thermoextrap_amd import engine as eng

    N, C, order, nrep = 1_000_000, 32, 4, 424
    g = torch.Generator(device="cuda").manual_seed(12)
    u = 174.85 + 5.31 * torch.randn(N, generator=g, dtype=torch.float64, device="cuda")
    x = 0.2 + 1e-3 * u[:, None] + 0.05 * torch.randn(N, C, generator=g, dtype=torch.float64, device="cuda")
    y = 0.5 * x + 1.0
    L = eng._L()
    assert L.txm_resample_kernel(N, C, nrep, order, -1, 1, 1) == (3 | 0x100)
    assert L.txm_resample_kernel(N, C, 40, order, -1, 1, 1) == 2          # what the 40-replicate slab would pick on its own
    s = eng.DeviceSampler(21, nrep, N, rep0=7)
    old = eng.WORKSPACE_BUDGET_BYTES
    prep = eng.ResamplePrep()
    try:
        eng.WORKSPACE_BUDGET_BYTES = 1 << 50
        whole = eng.resample_vals(x, u, order, sampler=s, y=y, prep=prep)
        assert (prep.hits, prep.misses) == (0, 1)
        eng.WORKSPACE_BUDGET_BYTES = L.txm_resample_vals_ws_bytes_opts(N, C, 128, order, 3, 1) + L.txm_resample_y_ws_bytes(N, C, 128) + 4096
        assert eng._slab_size(L, N, C, nrep, order, 3, True) == 128
        parts = eng.resample_vals(x, u, order, sampler=s, y=y, prep=prep)
        assert (prep.hits, prep.misses) == (4, 1)                           # the unslabbed call's block served the four slabs
    finally:
        eng.WORKSPACE_BUDGET_BYTES = old
    assert torch.equal(whole[0], parts[0]) and torch.equal(whole[1], parts[1])
