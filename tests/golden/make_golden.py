#!/usr/bin/env python
"""
tests/golden/make_golden.py -- regenerates the committed golden vectors.

Runs ONLY in the build container, where /root/reference is mounted.  Nothing in
tests/, smoke() or bench.py reads /root/reference at run time; they read the
files this script writes next to itself:

  kat_notebooks.json   numbers parsed out of the *stored cell outputs* of the
                       reference's seeded notebooks (examples/usage/basic/*.ipynb)
                       -- the reference's own results for cmomy-backed
                       reduce/resample/derivs/predict, 4-5 significant digits.
  idealgas_seed0.npz   the notebook inputs (x, u at N=1e5) regenerated with the
                       recipe of src/thermoextrap/idealgas.py:166-208,403-421 and
                       checked against the stored x[:3], x[-3:] before saving.
  fixture_legacy.npz   tests/conftest.py:15-28 FixtureData(100, 5, order=5, seed=0)
                       inputs and the outputs of the reference's runnable legacy
                       oracle (src/thermoextrap/legacy/utilities.py, extrap.py),
                       imported here by file path -- the same oracle the
                       reference's slow tests use (tests/test_beta.py:17-26,
                       42-47, 519-543, 654-676; tests/test_volume.py:56-74).
  lnpi_sample_data.json  byte copy of tests/lnpi_data/sample_data.json, the
                       reference's only committed numeric golden file
                       (tests/test_lnPi.py:106-159; NIST public-domain licence).

The full package cannot be imported (cmomy/xarray/numba are not installable
here: ordinary ModuleNotFoundError, no permission was denied), so vectors for the
cmomy-backed calls come from the notebook outputs, not from running them.
"""

from __future__ import annotations

import importlib
import json
import math
import re
import shutil
import sys
import types
from pathlib import Path

import numpy as np

REF = Path("/root/reference")
HERE = Path(__file__).resolve().parent


# ---------------------------------------------------------------------------
def load_legacy():
    """Import legacy/utilities.py and legacy/extrap.py by path under a private
    package name (legacy/__init__.py pulls in gpflow-dependent modules)."""
    pkg = types.ModuleType("txlegacy")
    pkg.__path__ = [str(REF / "src/thermoextrap/legacy")]
    sys.modules["txlegacy"] = pkg
    util = importlib.import_module("txlegacy.utilities")
    extrap = importlib.import_module("txlegacy.extrap")
    return util, extrap


def nb_outputs(path):
    d = json.loads(Path(path).read_text())
    out = {}
    for i, c in enumerate(d["cells"]):
        if c["cell_type"] != "code":
            continue
        texts = []
        for o in c.get("outputs", []):
            if "text" in o:
                texts.append("".join(o["text"]))
            elif "data" in o and "text/plain" in o["data"]:
                texts.append("".join(o["data"]["text/plain"]))
        out[i] = "\n".join(texts)
    return out


_num = re.compile(r"[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?")


def parse_array(text):
    """Numbers inside the first ``array(...)`` of an xarray repr."""
    m = re.search(r"array\((.*?)\)\s*(?:\n[A-Z]|$)", text, re.S)
    body = m.group(1) if m else text
    return [float(v) for v in _num.findall(body)]


def make_kat():
    base = REF / "examples/usage/basic"
    kat = {"source": "stored outputs of /root/reference/examples/usage/basic/*.ipynb @ v0.6.0"}

    do = nb_outputs(base / "Data_Organization.ipynb")
    kat["data_org"] = {
        "recipe": "rng=default_rng(0); positions from rng.random((100000,1000)), beta=5.6, vol=1; order=2",
        "x_head_tail": parse_array(do[13]),          # cell 13: data.xv
        "xave": parse_array(do[14]),                 # cell 14
        "u": parse_array(do[16]),                    # cell 16: raw <u^k>
        "du": parse_array(do[18]),                   # cell 18
        "xu": parse_array(do[20]),                   # cell 20
        "dxdu": parse_array(do[22]),                 # cell 22
        "values": parse_array(do[24]),               # cell 24: [2,3] state
        "resample_nrep3": parse_array(do[35]),       # cell 35: DataCentralMomentsVals.resample -> [3,2,3]
        "block_resample_nrep3": parse_array(do[41]),  # cell 41: DataCentralMoments.resample (100 blocks)
        "vec_values": parse_array(do[47]),           # cell 47: vals=(x, x^2) -> [2,2,3]
        "vec_resample_nrep3": parse_array(do[48]),   # cell 48 -> [3,2,2,3]
        "vec_block_reduce": parse_array(do[51]),     # cell 51
        "vec_block_resample_nrep3": parse_array(do[52]),  # cell 52
    }
    # first rows of the per-block states (cell 39), [100,2,3] truncated by the repr
    blk = [float(v) for v in _num.findall(do[39].split("array(")[1])][: 6 * 6]
    kat["data_org"]["block_values_first6"] = blk

    c1 = nb_outputs(base / "Temperature_Extrap_Case1.ipynb")
    t10 = c1[10]
    kat["case1"] = {
        "recipe": "same data; order=6; beta_ref=5.6; betas=arange(0.1,10,0.5)",
        "rng_random_after_data": float(_num.findall(c1[9])[0]),       # cell 9: 0.045
        "predict_beta0p1_order6": float(re.search(r"Extrapolation: ([-\d.]+)", t10).group(1)),
        "perturb_beta0p1": float(re.search(r"Perturbation:, ([-\d.]+)", t10).group(1)),
        "predict_betas4_order2": [float(v) for v in _num.findall(re.search(r"Extrapolation: \[(.*?)\]", t10).group(1))],
        "boot100_predict_mean": parse_array(c1[14]),                   # cell 14
        "boot100_predict_std": parse_array(c1[15]),                    # cell 15
        "derivs_N1e5": [float(v) for v in _num.findall(c1[17].split("With N_configs = 100000:")[1].split("]")[0])],
        "true_coefs": [float(v) for v in _num.findall(c1[17].split("True extrapolation coefficients:")[1].split("]")[0])],
    }

    for name, nbn in [("case2", "Temperature_Extrap_Case2.ipynb"), ("case3", "Temperature_Extrap_Case3.ipynb"),
                      ("case4", "Temperature_Extrap_Case4.ipynb"), ("custom", "Customized_Derivatives.ipynb")]:
        outs = nb_outputs(base / nbn)
        kat[name + "_raw_outputs"] = {str(k): v[:1500] for k, v in outs.items() if v.strip()}
    return kat


def idealgas_seed0(beta=5.6, vol=1.0):
    """idealgas.generate_data((100000, 1000), beta, vol) on default_rng(0)
    (src/thermoextrap/idealgas.py:166-189, 403-421)."""
    rng = np.random.default_rng(0)
    r = rng.random((100_000, 1000))
    pos = (-1.0 / beta) * np.log(1.0 - r * (1.0 - np.exp(-beta * vol)))
    x = pos.mean(axis=-1)
    u = pos.sum(axis=-1)
    # the generator state after the draw is what the notebooks' samplers continue from
    rng2 = np.random.default_rng(0)
    rng2.bit_generator.advance(100_000 * 1000)
    assert rng2.random() == rng.random(), "PCG64.advance does not reproduce the post-data state"
    return x, u


def make_fixture(util, extrap):
    """FixtureData(100, 5, order=5, seed=0): tests/conftest.py:15-28."""
    n, nv, order = 100, 5, 5
    rng = np.random.default_rng(0)
    u = rng.random(n)
    x = rng.random((n, nv))
    ub = rng.random(n)
    xb = rng.random((n, nv))
    out = dict(u=u, x=x, ub=ub, xb=xb, order=order, beta0=0.5)

    # raw moments: tests/test_data.py:7-38
    ufunc, xufunc = util.buildAvgFuncs(x, u, order)
    out["raw_u"] = np.array([ufunc(i) for i in range(order + 1)])
    out["raw_xu"] = np.array([xufunc(i) for i in range(order + 1)])

    # derivatives: tests/conftest.py:107-112, tests/test_beta.py:17-26
    fs = [util.symDerivAvgX(i) for i in range(order + 1)]
    out["derivs"] = np.array([fs[i](ufunc, xufunc) for i in range(order + 1)])

    # predict: tests/test_beta.py:42-47 (legacy ExtrapModel.predict, extrap.py:84-118)
    em = extrap.ExtrapModel(maxOrder=order)
    em.train(0.5, xData=x, uData=u, saveParams=True)
    out["predict_betas"] = np.array([0.3, 0.4])
    out["predict_order3"] = em.predict([0.3, 0.4], order=3)
    out["predict_order5"] = em.predict([0.3, 0.4], order=5)
    np.testing.assert_allclose(em.params, out["derivs"])

    # state B (weighted/interp tests use it): derivs only
    ufb, xufb = util.buildAvgFuncs(xb, ub, order)
    out["derivs_b"] = np.array([fs[i](ufb, xufb) for i in range(order + 1)])

    # -log<x>: tests/test_beta.py:483-516 (LogAvgExtrapModel; Faa di Bruno with sympy.bell)
    from sympy import bell

    dl = np.zeros((order + 1, nv))
    for o in range(order + 1):
        if o == 0:
            dl[o] = -np.log(xufunc(0))
            continue
        for k in range(1, o + 1):
            diffs = np.array([fs[v](ufunc, xufunc) for v in range(1, o - k + 2)])
            for v in range(nv):
                dl[o, v] += math.factorial(k - 1) * ((-1 / xufunc(0)[v]) ** k) * float(bell(o, k, diffs[:, v]))
    out["derivs_minus_log"] = dl

    # x depends on beta: tests/test_beta.py:617-676 (ExtrapModelDependent)
    rng_d = np.random.default_rng(1)
    xdep = rng_d.random((n, order + 1, nv))
    out["x_dep"] = xdep
    ufd, xufd = util.buildAvgFuncsDependent(xdep, u, order)
    fd = [util.symDerivAvgXdependent(i) for i in range(order + 1)]
    out["derivs_dep"] = np.array([fd[i](ufd, xufd) for i in range(order + 1)])
    out["raw_xu_dep"] = np.array([[xufd(d, k) for k in range(order + 1)] for d in range(order + 1)])  # [deriv, umom, val]

    # volume (ideal gas 1-D variant): tests/test_volume.py:17-53 VolumeExtrapModelIG.calcDerivVals
    vol = 1.0
    wT = np.array([u]).T
    x_ave = np.average(x, axis=0)
    dv = np.zeros((2, nv))
    dv[0] = x_ave
    dv[1] = (np.average(x * wT, axis=0) - x_ave * np.average(u)) / vol + x_ave / vol
    out["derivs_volume_ig"] = dv
    out["volume"] = vol
    return out


def main():
    if not REF.exists():
        raise SystemExit("needs /root/reference (build container only)")
    util, extrap = load_legacy()

    kat = make_kat()
    x, u = idealgas_seed0()
    ht = kat["data_org"]["x_head_tail"]
    got = np.r_[x[:3], x[-3:]]
    assert np.allclose(got, ht, atol=5e-5), (got, ht)
    (HERE / "kat_notebooks.json").write_text(json.dumps(kat, indent=1))
    np.savez_compressed(HERE / "idealgas_seed0.npz", x=x, u=u)
    # Customized_Derivatives.ipynb cell 8: the same draw at beta = 1, volume = 5 (the volume-extrapolation example; u = 1000 x)
    xv5, _ = idealgas_seed0(beta=1.0, vol=5.0)
    np.savez_compressed(HERE / "idealgas_seed0_vol5.npz", x=xv5)

    fx = make_fixture(util, extrap)
    np.savez_compressed(HERE / "fixture_legacy.npz", **fx)

    shutil.copyfile(REF / "tests/lnpi_data/sample_data.json", HERE / "lnpi_sample_data.json")
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
