"""The committed evidence under profiles/ is consistent with itself and with what the documents say about it.

* `roofline.traffic` of a bench line is the HBM traffic of the timed CALL -- the sum over the kernels the call launches
  (count-table generator, every contraction pass, finalize, ...) of the per-launch figures of the rocprofv3 --pmc summary it
  names -- recomputed here from that committed file (round-5 verdict item 2: the line carried one launch's 132 GB where the
  call moves 370).
* every `rNN?_...` file name DESIGN.md, README.md and profiles/README.md cite exists under profiles/.
"""

import json
import re
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
PROF = ROOT / "profiles"
sys.path.insert(0, str(ROOT))


def test_call_traffic_of_the_round5_summary_is_the_verdicts_370_GB():
    import bench

    d = json.loads((PROF / "r05f_traffic.json").read_text())
    total, per = bench.call_traffic(d["kernels"], "int8_table", 32)
    assert 3.69e11 < total < 3.72e11
    names = sorted(k.split("<")[0] for k in per)
    assert names.count("txm::resample_i8g_kernel") == 2 and "txm::count_table_kernel" in names and "txm::resample_finalize_i8_kernel" in names
    # a first call on a data object adds the pre-pass (one more read of the samples)
    cold, _ = bench.call_traffic(d["kernels"], "int8_table", 32, prepass=True)
    assert 2.6e10 < cold - total < 2.9e10
    # per 32-column group: a 64-column state launches the per-group kernels twice, the generator once
    two, per2 = bench.call_traffic(d["kernels"], "int8_table", 64)
    gen = per["txm::count_table_kernel"]
    assert abs((two - gen) - 2 * (total - gen - per["txm::i8_info_kernel"]) - per["txm::i8_info_kernel"]) < 1.0


@pytest.mark.parametrize("name", sorted(p.name for p in list(PROF.glob("r06*_bench.json")) + list(PROF.glob("r06*_bench_c4.json"))))
def test_bench_line_traffic_is_the_sum_over_the_calls_kernels(name):
    import bench

    rec = json.loads((PROF / name).read_text())
    r = rec["roofline"]
    if r.get("traffic") is None:
        pytest.skip("no PMC summary matched this line's kernel sources")
    src = re.match(r"profiles/(\S+)", r["traffic_source"]).group(1)
    d = json.loads((PROF / src).read_text())
    cfg = rec["config"]
    assert (d["workload"]["n_samp"], d["workload"]["n_obs"], d["workload"]["order"], d["workload"]["nrep"]) == \
        (cfg["n_samp"], cfg["n_obs"], cfg["order"], cfg["nrep"])
    total, per = bench.call_traffic(d["kernels"], rec["step_breakdown_ms"]["int8_kernel"], cfg["n_obs"],
                                    prepass=not rec["step_breakdown_ms"]["prepass_reused"], has_y="dx/dq" in cfg["workload"])
    assert total == pytest.approx(r["traffic"], rel=1e-12) and per == pytest.approx(r["kernels"], rel=1e-12)
    assert sum(r["kernels"].values()) == pytest.approx(r["traffic"], rel=1e-12)
    assert r["traffic_ratio"] == pytest.approx(r["traffic"] / r["algorithmic_bytes"], rel=1e-12)
    assert r["hbm_executed_GBs"] == pytest.approx(r["traffic"] / (r["ms"] * 1e-3) / 1e9, rel=1e-9)
    assert r["traffic"] > max(r["kernels"].values())          # more than any one launch


@pytest.mark.parametrize("name", sorted(p.name for p in PROF.glob("r06*_bench_c[25].json")))
def test_narrow_bench_lines_carry_the_calls_traffic_from_the_narrow_counters(name):
    """BASELINE configs 2 and 5 (round-5 verdict item 3: `traffic` was null for every narrow launch): the line's traffic is the
    sum over the kernels of ONE bootstrap call in the tools/narrow_pmc.sh summary it names, recomputed here from that file."""
    import bench

    rec = json.loads((PROF / name).read_text())
    r = rec["roofline"]
    if r.get("traffic") is None:
        pytest.skip("no narrow PMC summary matched this line's kernel sources (lines before the summary carried FETCH / WRITE of every kernel)")
    m = re.match(r"profiles/(\S+) \[(c[25])\]", r["traffic_source"])
    d = json.loads((PROF / m.group(1)).read_text())["configs"][m.group(2)]
    cfg = rec["config"]
    assert (d["workload"]["n_samp"], d["workload"]["n_obs"], d["workload"]["order"], d["workload"]["nrep"]) == \
        (cfg["n_samp"], cfg["n_obs"], cfg["order"], cfg["nrep"])
    total, per = bench.call_traffic(d["kernels_traffic"], d["call_kernel"], cfg["n_obs"])
    assert total == pytest.approx(r["traffic"], rel=1e-12) and total == pytest.approx(d["call_hbm_bytes"], rel=1e-12)
    assert per == pytest.approx(r["kernels"], rel=1e-12)
    assert r["traffic_ratio"] == pytest.approx(total / r["algorithmic_bytes"], rel=1e-12) and 1.0 < r["traffic_ratio"] < 20.0
    assert any("resample_i8" in k for k in per) and any("finalize" in k for k in per)


def test_no_document_cites_a_profile_file_that_does_not_exist():
    have = {p.name for p in PROF.iterdir()}
    missing = []
    for doc in (ROOT / "DESIGN.md", ROOT / "README.md", PROF / "README.md"):
        for tok in re.findall(r"r0[0-9][a-z]?_[A-Za-z0-9_.*{},|-]+", doc.read_text()):
            tok = tok.rstrip(".,;:")
            if "*" in tok or "{" in tok or "|" in tok:            # patterns (r05f_*, r05f_{a,b}.json): at least one match
                stem = re.split(r"[*{|]", tok)[0]
                if not any(h.startswith(stem) for h in have):
                    missing.append((doc.name, tok))
                continue
            if tok not in have and not any(h.startswith(tok) for h in have):
                missing.append((doc.name, tok))
    assert not missing, missing
