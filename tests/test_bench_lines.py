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


def test_bench_starts_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` with no launcher around it: the parent process starts N ranks under
    torch.distributed.run (rendezvous on 127.0.0.1), passes its own arguments on, returns the children's
    status -- and does so without importing torch or touching a GPU."""
    import subprocess
    import sys
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--config", "assoc"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--config", "assoc"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_bench_parent_does_not_load_torch():
    """The launching parent must never initialise a GPU: it does not even import torch."""
    import subprocess
    import sys
    code = ("import sys, subprocess; sys.argv=['bench.py','--gpus','2'];"
            "subprocess.call=lambda cmd, env=None: 0\n"
            "import bench\n"
            "try:\n    bench.main()\nexcept SystemExit as e:\n    assert e.code == 0\n"
            "assert 'torch' not in sys.modules, 'torch imported by the launching parent'\nprint('ok')")
    env = {k: v for k, v in os.environ.items() if k != "WORLD_SIZE"}
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr
