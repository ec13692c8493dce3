"""bench.py's output contract (one JSON line on rank 0) checked on a tiny workload."""
import json
import os
import subprocess
import sys

import pytest

from helpers import ROOT

gpu = pytest.mark.gpu


def _run(*extra):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--width", "96", "--height", "64", "--spp", "4",
                        "--steps", "2", "--warmup", "1", "--cpu-spp", "2", *extra], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


@gpu
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
    # the bound that binds is VALU issue; flat scalars only; counters come from a profile of the same kernel build or
    # are null (this tiny workload has no profile), never a stale constant; frac <= 1 whenever present
    assert rf["bound"] == "valu_issue" and rf["peak"] == 1228.8 and rf["hbm_peak_gbs"] == 8000.0
    assert all(not isinstance(v, (dict, list)) for v in rf.values())
    assert rf["kernel_ms_hip_events"] > 0 and "k_render" in rf["kernel"]
    if rf["frac"] is None:
        assert rf["achieved"] is None and rf["traffic"] is None and rf["counters"]
    else:
        assert 0 < rf["frac"] <= 1 and rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], abs=1e-3)
    # the CPU leg rendered the same seed at the same size and spp as one GPU frame: byte for byte, segment for segment
    assert d["config"]["parity_full_frame"] == "bit-exact", d["config"]["parity_full_frame_detail"]
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "Mrays/s" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb


@gpu
def test_forced_distributed_path_line():
    env = dict(os.environ, RTR_BENCH_FORCE_DISTRIBUTED="1", MASTER_PORT="29561")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--width", "96", "--height", "64", "--spp", "4",
                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 1 and "samples" in d["config"]["sharding"] and "cpu_baseline" not in d
    assert 96 * 64 * 4 <= d["config"]["segments_per_frame"] <= 96 * 64 * 4 * 6


def test_gpus_n_without_a_launcher_starts_n_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher must start two ranks itself (CPU stand-in of the ranks over gloo:
    RTR_BENCH_PLUMBING=1), split the samples between them, reduce, and print ONE line with n_gpus = 2."""
    env = dict(os.environ, RTR_BENCH_PLUMBING="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--spp", "13", "--steps", "2", "--warmup", "0"],
                       capture_output=True, text=True, cwd=ROOT, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["plumbing"] is True
    assert d["sample_ranges"] == [[0, 7], [7, 6]] and d["reduce_matches_single_rank"] is True


def test_gpus_n_refuses_when_fewer_devices_are_visible():
    """Never a silent fall-back to fewer GPUs: this container has none, so --gpus 2 must exit non-zero with a message
    and no JSON line (on a GPU box with enough devices the same command runs)."""
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible: the refusal cannot be provoked")
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RTR_BENCH_PLUMBING"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, cwd=ROOT,
                       env=env, timeout=300)
    assert r.returncode != 0 and "--gpus 2" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_world_size_mismatch_is_an_error():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True, cwd=ROOT,
                       env=env, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)
