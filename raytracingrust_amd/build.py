"""Build the gfx950 shared library (HIP kernels + C ABI) in-tree.

`python -m raytracingrust_amd.build` or `__graft_entry__.build()`.  hipcc cross-compiles for gfx950
without a GPU.  The .so is git-ignored but travels to the GPU box with the snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "_lib")
LIB_PATH = os.path.join(LIB_DIR, "librtr_hip.so")

SOURCES = ["rtr_api.cpp", "rtr_scene.cpp", "rtr_kernels.hip", "rtr_lbvh.hip", "rtr_sort.hip"]
HEADERS = ["rtr_device.h", "rtr_scene.h", os.path.join("..", "..", "include", "rtr_hip.h")]

# -ffp-contract=off + correctly rounded divide/sqrt: the kernels must round like the reference's f32 code
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math",
    # no SLP vectorisation: packed v_pk_{mul,add}_f32 issue at half rate on gfx950 and need v_mov shuffles to
    # pair Vec3 components -- measured -10 % kernel time and -8 VGPRs on the headline kernel without them
    "-fno-slp-vectorize",
    "-Wall", "-Wextra",
]


def kernel_sha() -> str:
    """Hash of everything that decides the device code: kernel sources, shared headers, launcher and compiler flags.
    rocprofv3 summaries under profiles/ carry it; bench.py reports their counters only while it still matches."""
    import hashlib

    h = hashlib.sha256()
    # every source and header of the library: the flattener (rtr_scene.*) and the device builder (rtr_lbvh.hip) decide the
    # tree layout the kernels walk -- leaf pairs, splits, stack depth -- and with it LDS per wave, occupancy and counters
    for name in SOURCES + HEADERS:
        with open(os.path.join(CSRC, name), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()[:16]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (needed to build librtr_hip.so for gfx950)")


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_variant(name: str, defines, verbose: bool = False) -> str:
    """Kernel experiments: the same sources with extra -D flags, written next to the product library."""
    out = os.path.join(LIB_DIR, f"librtr_hip_{name}.so")
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [_hipcc()] + FLAGS + [d if d.startswith("-") else f"-D{d}" for d in defines] + ["-o", out] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed")
    return out


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    tmp = LIB_PATH + ".tmp"
    cmd = [_hipcc()] + FLAGS + ["-o", tmp] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed building librtr_hip.so")
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "--variant":
        print(build_variant(sys.argv[2], sys.argv[3:], verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
