"""Dataset-valued observables, host side (no GPU): xrlite.Dataset and the stacking of its variables into one
(rec, column) matrix and back (thermoextrap_amd.data.stack_dataset / unstack_dataset; reference data.py:347-350)."""
import numpy as np
import pytest

from thermoextrap_amd.xrlite import DataArray, Dataset, as_dataset, is_dataset


def _ds(n=7):
    rng = np.random.default_rng(0)
    a = DataArray(rng.normal(size=(n, 3)), ("rec", "val"), coords={"val": [10, 20, 30]})
    b = DataArray(rng.normal(size=n), ("rec",))
    c = DataArray(rng.normal(size=(2, n, 4)), ("p", "rec", "q"))       # record dim in the middle
    return Dataset({"a": a, "b": b, "c": c})


def test_dataset_container():
    ds = _ds()
    assert is_dataset(ds) and not is_dataset(ds["a"]) and as_dataset(ds) is ds
    assert list(ds) == ["a", "b", "c"] and len(ds) == 3 and "b" in ds
    assert ds.sizes == {"rec": 7, "val": 3, "p": 2, "q": 4}
    assert ds["b"].name == "b"
    doubled = ds.map(lambda v: v * 2.0)
    np.testing.assert_array_equal(doubled["c"].values, 2.0 * ds["c"].values)
    with pytest.raises(TypeError):
        Dataset({"x": np.zeros(3)})
    with pytest.raises(ValueError):
        Dataset({"a": DataArray(np.zeros(3), "rec"), "b": DataArray(np.zeros(4), "rec")}).sizes


def test_stack_and_unstack_roundtrip():
    from thermoextrap_amd.data import DS_DIM, stack_dataset, unstack_dataset

    ds = _ds()
    mat, layout = stack_dataset(ds, "rec")
    assert mat.dims == ("rec", DS_DIM) and mat.shape == (7, 3 + 1 + 8)
    assert [(name, lo, hi) for name, lo, hi, *_ in layout] == [("a", 0, 3), ("b", 3, 4), ("c", 4, 12)]
    np.testing.assert_array_equal(mat.values[:, 3], ds["b"].values)
    np.testing.assert_array_equal(mat.values[:, 4:].reshape(7, 2, 4), np.moveaxis(ds["c"].values, 1, 0))
    # a result with the stacked axis anywhere (here: (order, rep, column)) splits back into the variables' own dims
    res = DataArray(np.arange(2 * 5 * 12, dtype=float).reshape(2, 5, 12), ("order", "rep", DS_DIM))
    out = unstack_dataset(res, layout)
    assert out["a"].dims == ("order", "rep", "val") and out["b"].dims == ("order", "rep")
    assert out["c"].dims == ("order", "rep", "p", "q") and out["c"].shape == (2, 5, 2, 4)
    assert out["a"].coords["val"].tolist() == [10, 20, 30]
    np.testing.assert_array_equal(out["b"].values, res.values[:, :, 3])
    np.testing.assert_array_equal(out["c"].values.reshape(2, 5, 8), res.values[:, :, 4:])
    with pytest.raises(ValueError):
        stack_dataset(Dataset({"z": DataArray(np.zeros(3), ("other",))}), "rec")
    with pytest.raises(ValueError):
        stack_dataset(Dataset({"a": DataArray(np.zeros(3), "rec"), "b": DataArray(np.zeros(4), "rec")}), "rec")
    with pytest.raises(ValueError):
        stack_dataset(Dataset({}), "rec")
