"""bench.py's stdout contract: the LAST line is one compact JSON object the driver can parse from a bounded tail of stdout.
Round 4's line was 25 KB (nested other_configs, prose `sample` strings) and the driver recorded `parsed: null`; the formatter is
now checked on that very record (profiles/r04_bench_default.json, a full bench record) and on synthetic worst cases."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _canned():
    return json.load(open(os.path.join(ROOT, "profiles", "r04_bench_default.json")))


def _strings(o):
    if isinstance(o, dict):
        for v in o.values():
            yield from _strings(v)
    elif isinstance(o, str):
        yield o


def test_compact_line_fits_and_round_trips(bench):
    full = _canned()
    assert len(json.dumps(full)) > 20000  # the record that could not be parsed
    line = json.dumps(bench.compact_line(full, "bench_full.json"), separators=(",", ":"))
    assert len(line) < 4096 and len(line) <= bench.LINE_BUDGET and "\n" not in line
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["config"]["workload"].startswith("EVM circuit, 2^18")
    assert d["config"]["witness_copies"] == 3 and d["config"]["rw_rows"] == full["config"]["rw_rows"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert abs(r["frac"] - full["roofline"]["frac"]) < 1e-5 and r["traffic"] is not None
    for k in ("traffic_over_algorithmic", "algorithmic_bytes", "kernel", "kernel_ms", "open_ms", "pass_kernel_ms"):
        assert k in r, k
    assert abs(d["value"] - full["value"]) / full["value"] < 1e-5
    # <= 3 scalars per other configuration
    assert set(r["other_configs"]) == set(full["other_configs"])
    assert all(len(v) <= 5 and all(isinstance(x, (int, float)) for k, x in v.items() if k != "bound") for v in r["other_configs"].values())
    # every fraction says what it is a fraction OF: the headline's is algorithmic, the physical one sits beside it
    assert r["frac_bound"] == "hbm-algorithmic" and 0 < r["physical_frac"] < 1
    assert abs(r["physical_frac"] - r["traffic"] / (r["kernel_ms"] * 1e-3) / 1e9 / r["peak"]) < 1e-4
    for k, v in r["other_configs"].items():
        assert ("frac" in v) == ("bound" in v), k
        if "bound" in v:
            assert v["bound"] in ("hbm-physical", "hbm-algorithmic", "valu"), (k, v)
    assert r["other_configs"]["tx_2p14"]["bound"] == "valu" and r["other_configs"]["state_2p20"]["bound"] == "hbm-physical"
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and isinstance(c["value"], float) and c["cores"] == 1 and len(c["sample"]) <= 120
    assert set(c["legs"]) == set(full["cpu_baseline"]["legs"])
    assert all(set(v) == {"value", "cores"} for v in c["legs"].values())  # numbers only
    assert max(len(s) for s in _strings(d)) <= 150  # no prose anywhere but the few short labels


def test_compact_line_survives_bloat(bench):
    """a record with far more side blocks than any run produces still yields a line under the budget"""
    full = _canned()
    full["other_configs"] = {f"cfg_{i}": dict(full["other_configs"]["state_2p16"]) for i in range(40)}
    full["config"]["workload"] = "x" * 5000
    full["roofline"]["kernel"] = "k" * 5000
    full["cpu_baseline"]["sample"] = "s" * 5000
    line = json.dumps(bench.compact_line(full, "bench_full.json"), separators=(",", ":"))
    assert len(line) <= bench.LINE_BUDGET
    assert json.loads(line)["roofline"]["frac"] > 0


def test_compact_line_pass_workloads(bench):
    """the State / Tx / Super lines (nested roofline blocks in the full record) flatten too"""
    full = _canned()
    seen = 0
    for key, blk in full["other_configs"].items():
        if "roofline" not in blk:
            continue
        seen += 1
        rec = dict(blk, metric="BN254 constraint-rows/sec", n_gpus=1, higher_is_better=True, scaling="weak", vs_baseline=None, dtype="u256", data="synthetic",
                   config=dict(blk.get("config") or {}, workload=blk["workload"]))
        line = json.dumps(bench.compact_line(rec), separators=(",", ":"))
        assert len(line) <= bench.LINE_BUDGET, key
        d = json.loads(line)
        assert d["roofline"].get("frac") is not None and d["roofline"].get("kernel"), key
        assert all(not isinstance(v, dict) or k in ("other_configs", "per_circuit_kernel_ms") for k, v in d["roofline"].items()), key
    assert seen >= 4


def test_effective_cores(bench):
    n, how = bench.legs.effective_cores()
    assert 1 <= n <= (os.cpu_count() or 1) and how
