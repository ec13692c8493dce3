"""The C++ host front-end (host/rtr.hpp + host/rtr_render.cpp) over the same C ABI: it must lower scenes exactly
as the Python mirror does (CPU check, host-only context) and render the same frame (GPU check)."""
import os
import subprocess

import numpy as np
import pytest

import pyoracle
import raytracingrust_amd as rt
from helpers import ROOT, benchmark_scene, house_scene, scene_path, simple_scene

CLI = os.path.join(ROOT, "host", "rtr_render")


@pytest.fixture(scope="module")
def cli():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "host")], check=True, capture_output=True)
    assert os.path.exists(CLI)
    return CLI


def _info(cli, *args):
    out = subprocess.run([cli, *args, "--host-only", "--info"], check=True, capture_output=True, text=True, cwd=ROOT).stdout
    toks = out.split()
    return {toks[i]: int(toks[i + 1]) for i in range(0, len(toks), 2)}


def _py_info(scene):
    with rt.Renderer(-1) as r:
        r.upload_scene(scene)
        i = r.scene_info()
    return {"objects": i.n_objects, "spheres": i.n_spheres, "meshes": i.n_meshes, "triangles": i.n_triangles,
            "materials": i.n_materials, "leaves": i.n_leaves, "nodes": i.n_nodes, "height": i.tree_height,
            "spheres_only": i.spheres_only}


def test_cpp_lowering_matches_python_lowering(cli):
    assert _info(cli, "--scene", scene_path("simple.json")) == _py_info(simple_scene(1))
    assert _info(cli, "--scene", scene_path("benchmark.json")) == _py_info(benchmark_scene(1))
    # src/gui.rs:56-119 built in C++ (8 lowered materials: 6 + the mix's two children)
    assert _info(cli, "--preset", "gui") == _py_info(house_scene(1))


def test_cpp_reader_errors_are_reported_not_crashes(cli, tmp_path):
    bad = tmp_path / "bad.json"
    bad.write_text('{"camera": {}}')
    r = subprocess.run([cli, "--scene", str(bad), "--host-only"], capture_output=True, text=True)
    assert r.returncode == 1 and "missing field" in r.stderr
    r = subprocess.run([cli, "--scene", str(tmp_path / "absent.json"), "--host-only"], capture_output=True, text=True)
    assert r.returncode == 1 and "cannot open" in r.stderr


@pytest.mark.gpu
def test_cpp_front_end_renders_the_same_frame(cli, tmp_path, gpu_renderer):
    w, h, spp, seed = 64, 40, 4, 5
    ppm, raw = str(tmp_path / "f.ppm"), str(tmp_path / "f.bin")
    subprocess.run([cli, "--scene", scene_path("benchmark.json"), "--width", str(w), "--height", str(h), "--spp", str(spp),
                    "--seed", str(seed), "--out", ppm, "--f32", raw], check=True, capture_output=True, cwd=ROOT)
    data = open(ppm, "rb").read()
    header = f"P6\n{w} {h}\n255\n".encode()
    assert data.startswith(header)
    rgb = np.frombuffer(data[len(header):], dtype=np.uint8).reshape(h, w, 3)
    mean = np.fromfile(raw, dtype=np.float32).reshape(h, w, 3)
    scene = benchmark_scene(spp)
    o_rgba, o_mean, _ = pyoracle.OracleScene(scene).render(w, h, seed, threads=0)
    assert np.array_equal(rgb, o_rgba[..., :3])
    assert np.array_equal(mean.view(np.uint32), o_mean.view(np.uint32))


@pytest.mark.gpu
def test_cpp_gui_scene_matches_python_path(cli, tmp_path, gpu_renderer):
    w, h, spp, seed = 96, 54, 2, 9
    raw = str(tmp_path / "g.bin")
    subprocess.run([cli, "--preset", "gui", "--width", str(w), "--height", str(h), "--spp", str(spp), "--seed", str(seed),
                    "--f32", raw], check=True, capture_output=True, cwd=ROOT)
    mean = np.fromfile(raw, dtype=np.float32).reshape(h, w, 3)
    scene = house_scene(spp)
    gpu_renderer.upload_scene(scene)
    _, py_mean = gpu_renderer.render(w, h, seed=seed, want_rgba8=False, want_mean=True)
    assert np.array_equal(mean.view(np.uint32), py_mean.view(np.uint32))


@pytest.mark.gpu
def test_cpp_front_end_over_several_ranks(cli, tmp_path):
    """`rtr_render --devices 0,0`: rtr::Renderer(std::vector<int>) = rtr_create_multi from C++ (two ranks on the one GPU
    of this box: the library adds the partial sums itself there).  Sample split: within float addition order of the
    single-context frame; `--devices 0` (one rank through RCCL): that frame bit for bit."""
    w, h, spp, seed = 64, 40, 6, 5
    raws = {}
    for name, extra in (("single", ["--device", "0"]), ("one_rank", ["--devices", "0"]), ("two_ranks", ["--devices", "0,0"])):
        raw = str(tmp_path / (name + ".bin"))
        subprocess.run([cli, "--scene", scene_path("simple.json"), "--width", str(w), "--height", str(h), "--spp", str(spp),
                        "--seed", str(seed), "--f32", raw, *extra], check=True, capture_output=True, cwd=ROOT)
        raws[name] = np.fromfile(raw, dtype=np.float32).reshape(h, w, 3)
    assert np.array_equal(raws["one_rank"].view(np.uint32), raws["single"].view(np.uint32))
    rel = float(np.linalg.norm(raws["two_ranks"].astype(np.float64) - raws["single"]) / np.linalg.norm(raws["single"].astype(np.float64)))
    assert rel <= 1e-6, rel
