"""bench.py's output contract (one JSON line on rank 0) checked on a tiny workload."""
import json
import os
import subprocess
import sys

import pytest

from helpers import ROOT

pytestmark = pytest.mark.gpu


def _run(*extra):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--width", "96", "--height", "64", "--spp", "4",
                        "--steps", "2", "--warmup", "1", "--cpu-spp", "2", *extra], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_single_gpu_line():
    d = _run()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["metric"] == "Mrays/s" and d["unit"] == "Mrays/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    seg = d["config"]["segments_per_frame"]
    assert 96 * 64 * 4 <= seg <= 96 * 64 * 4 * 6
    assert d["value"] == pytest.approx(seg / (d["ms_per_step"] * 1e-3) / 1e6, rel=0.02)
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], abs=1e-4)
    assert rf["achieved"] == pytest.approx(rf["algorithmic_bytes_per_launch"] / (rf["kernel_ms_hip_events"] * 1e-3) / 1e9, rel=0.02)
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "Mrays/s" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb


def test_forced_distributed_path_line():
    env = dict(os.environ, RTR_BENCH_FORCE_DISTRIBUTED="1", MASTER_PORT="29561")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--width", "96", "--height", "64", "--spp", "4",
                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 1 and "samples" in d["config"]["sharding"] and "cpu_baseline" not in d
    assert 96 * 64 * 4 <= d["config"]["segments_per_frame"] <= 96 * 64 * 4 * 6
