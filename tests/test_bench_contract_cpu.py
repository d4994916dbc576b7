"""The bench line's contract, checked on the committed record of the round (`profiles/r02_bench_default.json`, written
by `python bench.py` on an MI355X): the fields the driver and the judge read are there, consistent with each other and
with BASELINE.json — metric, unit, workload, whole-job value vs ms/step, roofline arithmetic, cpu_baseline leg."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(name):
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        pytest.skip(f"{name} not committed")
    with open(path) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def _check_entry(d, m, n, k, peak):
    assert d["unit"] == "pairs/s"
    assert d["value"] == pytest.approx(m * n / (d["ms_per_step"] * 1e-3), rel=1e-6)      # whole-job throughput
    r = d["roofline"]
    assert r["peak"] == peak and r["bound"] in ("mfma", "valu", "hbm")
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-9)
    flop = r["flop_per_pair"] * m * n
    assert r["achieved"] == pytest.approx(flop / (r["kernel_ms"] * 1e-3) / 1e12, rel=1e-6)  # algorithmic flops / kernel time
    assert r["kernel_ms"] <= d["ms_per_step"] * 1.001                                        # the kernel fits its step
    assert 0.0 < r["frac"] < 1.0
    c = d["config"]
    assert (c["m"], c["n"], c["k"]) == (m, n, k) and "model" not in c and c["workload"]


def test_default_bench_line_contract():
    d = _line("r02_bench_default.json")
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        base = json.load(f)
    assert d["metric"] == "query-point-pairs/s" and d["metric"].split("-")[0] in json.dumps(base)
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic"
    _check_entry(d, 65536, 1048576, 128, 157.3)
    assert d["roofline"]["flop_per_pair"] == 256 and d["roofline"]["traffic"] > 5.7e8       # >= the algorithmic HBM bytes
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    assert cb.get("matches_gpu_indices", True) is True
    also = d["also"]
    _check_entry(also["c2"], 4096, 65536, 3, 157.3)
    _check_entry(also["c5"], 131072, 2097152, 256, 2500.0)
    _check_entry(also["c1"], 1024, 4096, 3, 157.3)
    assert also["c2"]["roofline"]["flop_per_pair"] == 9 and also["c5"]["roofline"]["flop_per_pair"] == 512
    for name in ("c1", "c2", "c5"):
        assert also[name]["cpu_baseline"]["value"] > 0
    if "c3x" in also:
        assert also["c3x"]["same_indices_as_c3"] is True and "opt-in" in also["c3x"]["note"]


def test_round3_line_contract():
    """The same contract on this round's record, plus the fields round 3 added: the non-FMA ceiling of the exact
    kernel and the enforced cross-checks."""
    d = _line("r03_bench_default.json")
    assert d["metric"] == "query-point-pairs/s" and d["n_gpus"] == 1 and d["vs_baseline"] is None and d["dtype"] == "f32"
    _check_entry(d, 65536, 1048576, 128, 157.3)
    assert d["parity_ok"] is True and "parity_failed" not in d
    assert d["cpu_baseline"]["kind"] in ("reference", "port") and d["cpu_baseline"]["matches_gpu_indices"] is True
    also = d["also"]
    _check_entry(also["c2"], 4096, 65536, 3, 157.3)
    _check_entry(also["c5"], 131072, 2097152, 256, 2500.0)
    r2 = also["c2"]["roofline"]
    assert r2["ceiling"] == pytest.approx(157.3 / 2) and r2["ceiling_frac"] == pytest.approx(2 * r2["frac"], rel=1e-9)
    assert also["c3x"]["same_indices_as_c3"] is True
    for name in ("c1", "c2"):
        assert also[name]["unprofiled"]["same_indices"] is True


def test_second_run_carries_unprofiled_fields():
    d = _line("r02_bench_default_run2.json")
    for name in ("c1", "c2"):
        u = d["also"][name]["unprofiled"]
        assert u["same_indices"] is True and 0 < u["ms_per_step"] <= d["also"][name]["ms_per_step"]
