"""The line bench.py prints LAST must fit the driver's record (8 KB of stdout are kept; round 3's 20 KB line was lost).

The compact line is built from a canned full record — the committed detail of an earlier lease — so this runs without a GPU."""

import glob
import json
import os

import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "config", "roofline", "cpu_baseline")


def _canned():
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_bench_detail.json"))) or [os.path.join(ROOT, "profiles", "r03_bench.json")]
    with open(paths[-1]) as fh:
        return json.load(fh)


def test_compact_line_fits_and_keeps_the_contract():
    detail = _canned()
    line = bench.compact_line(detail, "gpurun_out/bench_detail.json")
    text = json.dumps(line)
    assert len(text) < 6144, len(text)   # (the driver parsed round 5's 3.8 KB line whole and keeps its last 2000 characters verbatim)
    for key in REQUIRED:
        assert key in line, key
    assert line["value"] == pytest.approx(detail["value"], rel=1e-6)
    assert set(("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(line["roofline"])
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(line["cpu_baseline"])
    assert line["config"]["workload"] and "model" not in line["config"]
    assert line["secondary"]["roofline"]["frac"] is not None and line["secondary"]["cpu_baseline"]["value"] is not None
    assert list(line)[-1] == "secondary" and list(line)[-2] == "legs"   # what the round is judged on sits in the record's verbatim tail
    assert json.loads(text) == line  # one JSON object, no NaN/Infinity tokens


def test_compact_line_is_bounded_whatever_the_detail_holds():
    """Every free-text field is clipped: notes of any length, extra legs and extra keys cannot push the line past the record."""
    detail = _canned()
    blob = "x" * 20000

    def bloat(obj):
        if isinstance(obj, dict):
            for k in list(obj):
                if isinstance(obj[k], str):
                    obj[k] = obj[k] + blob
                else:
                    bloat(obj[k])
            obj["extra_note"] = blob
        elif isinstance(obj, list):
            for v in obj:
                bloat(v)

    bloat(detail)
    detail["legs"]["an_extra_leg"] = {"value": 1.0, "note": blob}
    detail["emulated_ranks"] = {"ranks": 8, "total_perms": 100000, "shard_seconds": [0.012] * 8, "one_gpu_seconds": 0.095, "PROJECTION": blob}
    text = json.dumps(bench.compact_line(detail, "gpurun_out/bench_detail.json"))
    assert len(text) < 8000, len(text)


def test_compact_line_survives_a_bare_record():
    """N > 1 ranks print no CPU baseline and no legs; skipped legs leave no key behind."""
    line = bench.compact_line({"metric": "m", "value": 1.0, "unit": "u", "n_gpus": 8, "steps": 1, "warmup": 0, "ms_per_step": 1.0,
                               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
                               "config": {"workload": "w"}, "roofline": {"kernel": "k", "bound": "hbm", "frac": None}}, None)
    assert line["n_gpus"] == 8 and "cpu_baseline" not in line and "legs" not in line and "secondary" not in line
    assert len(json.dumps(line)) < 2000
