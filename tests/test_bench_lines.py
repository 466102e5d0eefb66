"""The bench lines kept under profiles/ (the newest of each configuration) carry the fields the bench
contract names: metric, value, unit, n_gpus, steps, warmup, ms_per_step, higher_is_better, scaling,
vs_baseline, dtype, data, config.workload, a roofline block and a CPU baseline."""
import glob
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def newest(config):
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_bench_{config}.json")), key=os.path.basename)
    assert paths, config
    with open(paths[-1]) as fh:
        return json.load(fh), os.path.basename(paths[-1])


@pytest.mark.parametrize("config", ["loop", "batched", "assoc"])
def test_committed_bench_lines_follow_the_contract(config):
    d, name = newest(config)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, (name, key)
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and isinstance(d["config"]["workload"], str)
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["steps"] > 0
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, (name, key)
    assert 0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, (name, key)
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0
    if config == "loop":
        # whole-job rate and step time describe the same run; the kernel cannot take longer than a step
        units = d["config"]["correspondences"] * d["config"]["lm_iterations"]
        assert abs(d["value"] - units / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
        assert r["mean_launch_ms"] <= d["ms_per_step"] * 1.02
        assert r["bytes_per_launch"] == 80 * units
