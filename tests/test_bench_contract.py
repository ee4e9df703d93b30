"""CPU: the committed bench lines under profiles/ keep the contract of the task statement (one JSON object with the metric of
BASELINE.json, `roofline` and `cpu_baseline` objects), their numbers are consistent with each other, and the band tiling
`bench.py` picks per rank count divides the field in whole rows of threshold tiles."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(name):
    with open(os.path.join(ROOT, "profiles", name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def _bench_module():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)  # imports only the standard library and numpy at module level
    return mod


def _rounds():
    import glob
    import re

    return sorted({re.search(r"(r\d\d)_cfg3_bench", f).group(1) for f in glob.glob(os.path.join(ROOT, "profiles", "r??_cfg3_bench.json"))
                   if re.search(r"(r\d\d)_cfg3_bench", f).group(1) >= "r03"})


@pytest.mark.parametrize("rnd", _rounds())
def test_headline_line_has_the_contract_keys_and_consistent_numbers(rnd):
    d = _line(f"{rnd}_cfg3_bench.json")
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert "failed" not in d
    assert d["unit"].replace("*", "·") in base["metric"].replace("*", "·") or "Mcells" in d["unit"]
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "strong" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    # whole-job throughput = cells x input timesteps / step time
    cells, T = 720 * 1440, 36500
    assert d["value"] == pytest.approx(cells * T / (d["ms_per_step"] * 1e-3) / 1e6, rel=1e-6)
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-9)
    assert r["achieved"] == pytest.approx(r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9, rel=1e-6)
    assert r["traffic"] is None or r["traffic"] >= 0.9 * r["algorithmic_bytes_per_launch"]
    pr = d["pipeline_roofline"]  # the north-star fraction: whole-path bytes / step time
    assert pr["achieved"] == pytest.approx(pr["algorithmic_bytes_per_step_per_gpu"] / (d["ms_per_step"] * 1e-3) / 1e9, rel=1e-6)
    assert pr["frac"] == pytest.approx(pr["achieved"] / pr["peak"], rel=1e-9) and pr["frac"] < r["frac"]
    if rnd >= "r04":  # listed before the single-kernel figure, so that nobody reads the latter as "the target is met"
        keys = list(d)
        assert keys.index("pipeline_roofline") < keys.index("roofline")
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    s = d["config"]["summary"]
    assert d["extra"]["single_stream"]["n_extreme"] == s["n_extreme"]  # the same field, the same answer, whatever the schedule
    assert s.get("thr_unresolved", 0) == 0 and s["invalid_total"] == 0


def test_every_kept_line_of_the_100yr_field_gives_the_same_counts():
    """Tilings (6 bands x 2 streams, single stream, two ranks on one card), rounds and kernel generations: the field is the same
    synthetic field, so every committed cfg3 line must report the same ocean cells and the same number of extremes (derived by
    comparing the lines with each other, not pinned to a literal)."""
    import glob

    lines = []
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[3-9]_cfg3*bench*.json"))):
        d = _line(os.path.basename(f))
        if "summary" in d.get("config", {}):
            lines.append((os.path.basename(f), d["config"]["summary"]["n_ocean"], d["config"]["summary"]["n_extreme"], d["n_gpus"]))
    assert len(lines) >= 2
    assert len({(a, b) for _, a, b, _ in lines}) == 1, lines


@pytest.mark.parametrize("name", [n for r in _rounds() for n in (f"{r}_cfg2_bench.json", f"{r}_cfg4_bench.json", f"{r}_cfg5_bench.json",
                                                                  f"{r}_cfg3_single_stream_bench.json")])
def test_other_lines_parse_and_price_their_dominant_kernel(name):
    if not os.path.exists(os.path.join(ROOT, "profiles", name)):
        pytest.skip(f"{name} not kept this round")
    d = _line(name)
    r = d["roofline"]
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-9) and 0 < r["frac"] < 1
    assert d["ms_per_step"] > 0 and d["value"] > 0 and "workload" in d["config"] and "failed" not in d


def test_band_count_tiles_the_field_in_whole_tile_rows():
    b = _bench_module()
    assert [b.band_count(n) for n in (1, 2, 3, 4, 6, 8, 12, 24)] == [6, 6, 6, 8, 6, 8, 12, 24]
    assert b.band_count(5) == 0 and b.band_count(7) == 0
    for n in (1, 2, 3, 4, 6, 8, 12, 24):
        nb = b.band_count(n)
        assert nb % n == 0 and 720 % nb == 0 and (720 // nb) % 30 == 0
