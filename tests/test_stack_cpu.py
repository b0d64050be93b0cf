"""Host-side stacking of replicate derivatives (reference tests/test_stack.py:30-70),
on synthetic derivative arrays so that no GPU is needed."""

import numpy as np
import pytest

from thermoextrap_amd import stack
from thermoextrap_amd.xrlite import DataArray, assert_allclose


@pytest.fixture
def derivs():
    rng = np.random.default_rng(0)
    dims = ("beta", "order", "rep", "pair", "position")
    return DataArray(rng.random((2, 4, 3, 2, 4)), dims, coords={"beta": [0.1, 10.0], "position": np.linspace(0, 2, 4)})


def test_mean_var(derivs):
    x = derivs.isel(beta=0)
    out = stack.to_mean_var(x, dim="rep")
    assert_allclose(out.sel(stats="mean", drop=True), x.mean("rep"))
    assert_allclose(out.sel(stats="var", drop=True), x.var("rep"))
    out = stack.to_mean_var(x, dim="rep", concat_dim="var")
    assert_allclose(out.sel(var=0, drop=True), x.mean("rep"))
    assert_allclose(out.sel(var=1, drop=True), x.var("rep"))
    both = stack.apply_reduction(x, "rep", ["mean", lambda a, dim: a.var(dim)], concat_dim="stats")
    np.testing.assert_allclose(both.values, out.values)
    assert isinstance(stack.apply_reduction(x, "rep", ["mean", "var"]), list)


def test_stack(derivs):
    y_unstack = stack.to_mean_var(derivs, "rep")
    y_data = stack.stack_dataarray(y_unstack, x_dims=["beta", "order"], stats_dim="stats")
    assert y_data.dims == ("xstack", "ystack", "stats")
    x_data = stack.multiindex_to_array(y_data.indexes["xstack"])
    ij = 0
    for beta in y_unstack["beta"].values:
        for order in range(y_unstack.sizes["order"]):
            np.testing.assert_allclose((beta, order), x_data[ij, :])
            ij += 1
    y_test = y_unstack.transpose("beta", "order", ..., "stats")
    newshape = (y_test.sizes["beta"] * y_test.sizes["order"], -1, y_test.sizes["stats"])
    np.testing.assert_allclose(y_test.values.reshape(newshape), y_data.values)
    with pytest.raises(ValueError):
        stack.stack_dataarray(y_data, x_dims=["stats"])
    with pytest.raises(ValueError):
        stack.stack_dataarray(y_unstack, x_dims=["beta", "order"], stats_dim="stats", policy="raise")
    w = stack.wrap_like_dataarray(np.zeros(y_unstack.shape), y_unstack)
    assert w.dims == y_unstack.dims and set(w.coords) == set(y_unstack.coords)


def test_stacked_derivatives(derivs):
    sd = stack.StackedDerivatives.from_derivs(derivs, x_dims=["beta", "order"])
    assert sd.order == 3 and sd.alpha_name == "beta" and sd.order_dim == "order"
    x, y = sd.array_data(order=2)
    assert x.shape == (2 * 3, 2) and len(y) == 2 * 4 and y[0].shape == (6, 2)
    np.testing.assert_allclose(y[0][:, 0], derivs.mean("rep").isel(pair=0, position=0, order=slice(None, 3)).values.reshape(-1))
    np.testing.assert_allclose(y[5][:, 1], derivs.var("rep").isel(pair=1, position=1, order=slice(None, 3)).values.reshape(-1))
    assert sd.xindexer_from_arrays(beta=[0.1, 10.0]) == [(0.1, 0), (10.0, 0)]
    sd2 = stack.StackedDerivatives.from_mean_var(derivs.mean("rep"), derivs.var("rep"), x_dims=["beta", "order"])
    np.testing.assert_allclose(sd2.stacked().values, sd.stacked().values)
