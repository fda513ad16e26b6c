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
    for pat in ("bench_v1[0-9].json", "bench_r3[a-z].json"):
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", pat))):
            out.append((os.path.basename(f), json.loads(open(f).read().strip().splitlines()[-1])))
    return out


def test_round3_lines_carry_every_baseline_config():
    """configs[2..4], the single-pair latency and the f32-only rate are part of the ONE line, each
    all-pairs / PnP entry with its own roofline and its pair count; traffic comes from a committed PMC summary."""
    lines = [d for name, d in _lines() if name.startswith("bench_r3") and "other_configs" in d]
    assert lines, "no committed round-3 bench line with other_configs"
    d = lines[-1]
    oc = d["other_configs"]
    keys = list(oc)
    for want in ("configs[2]", "configs[3] cfg4 binary", "configs[3] cfg4 f32", "configs[4]", "configs[1] literally", "f32_only_matcher"):
        assert any(k.startswith(want) for k in keys), want
    for k, v in oc.items():
        if k.startswith(("configs[2]", "configs[3]")):
            assert v["pairs"] >= (4950 if k.startswith("configs[2]") else 100000) and v["hypotheses"] >= 4096
            assert v["ok_within_1deg"] > 0.99 and 0 < v["roofline"]["frac"] < 1
        if k.startswith("configs[4]"):
            assert [l["threshold_px"] for l in v] == [8.0, 2.0] and all(l["hypotheses"] >= 16384 and l["roofline"]["bound"] == "valu" for l in v)
        if k.startswith("configs[1]"):
            assert [l["pairs_per_batch"] for l in v] == [1, 8] and all(l["graph_table_equals_eager"] for l in v)
    r = d["roofline"]
    if r["traffic"] is not None:
        assert "profiles/r3/" in r["traffic_source"] and os.path.exists(os.path.join(ROOT, r["traffic_source"].split(":")[0]))
        assert 0.5 < r["traffic"] / (8.49e6 * d["config"]["pairs_per_step_per_gpu"]) < 1.5


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
