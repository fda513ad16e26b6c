"""The bench line's contract, checked on the committed lines under profiles/ (bench.py itself needs a GPU):
every key the driver reads, the roofline and cpu_baseline objects, and the values' sanity."""
import glob
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline"}


def _lines():
    out = []
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "bench_v1[0-9].json"))):
        out.append((os.path.basename(f), json.loads(open(f).read().strip().splitlines()[-1])))
    return out


def test_committed_bench_lines_follow_the_contract():
    lines = _lines()
    assert lines, "no committed bench line"
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for name, d in lines:
        assert KEYS <= set(d), (name, KEYS - set(d))
        assert d["unit"] == "pairs/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
        assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
        assert "model" not in d["config"]
        assert abs(d["value"] - d["n_gpus"] * d["config"]["pairs_per_step_per_gpu"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
        r = d["roofline"]
        assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(r)
        assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
        if "cpu_baseline" in d:
            c = d["cpu_baseline"]
            assert {"value", "unit", "cores", "kind", "sample"} <= set(c) and c["kind"] in ("port", "reference")
            assert c["unit"] == d["unit"] and c["value"] > 0
    if isinstance(base, dict) and "metric" in base:
        words = [w for w in ("pairs", "pose") if w in str(base["metric"]).lower()]
        assert all(w in lines[-1][1]["metric"].lower() for w in words)


def test_bench_source_keeps_the_flags_of_the_contract():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for flag in ("--gpus", "--steps", "--warmup"):
        assert f'"{flag}"' in src
    assert "LOCAL_RANK" in src and "WORLD_SIZE" in src and "127.0.0.1" in src
