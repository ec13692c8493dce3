#!/usr/bin/env python3
"""Diagnostic: wave-cycle shares of the render loop's sections from a -DRTR_X_STAMPS build (never the product).
usage (on the GPU box): python scripts/stamps.py [config] [spp]"""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RTR_HIP_LIBRARY"] = os.path.join(ROOT, "raytracingrust_amd", "_lib", "librtr_hip_stamps.so")
import raytracingrust_amd as rt  # noqa: E402
from raytracingrust_amd.configs import baseline_config  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
scene, W, H = baseline_config(cfg, spp)
r = rt.Renderer(0)
r.upload_scene(scene)
one = copy.copy(scene)
one.settings = copy.copy(scene.settings)
one.settings.samples_per_pixel = 1
for _ in range(5):
    r.render(8, 8, seed=1, scene=one)  # launches 0..4: ring slots the stamped launch writes below itself
r.render(W, H, seed=1)
ring = [int(v) for v in r.segment_counts(64)]
st = r.stats()
# (shade_segment is one section since it became a function; the ball-rejection stamp inside it went away: always 0)
names = ["loop top/regen tail", "after done block", "(unused)", "shade_segment: material fetch, sampler, scatter", "world.hit"]
tot = sum(ring[:5])
print(f"{cfg} {W}x{H}x{spp}: segments {ring[5]} kernel_ms {st.kernel_ms:.3f} (stamped build; read the shares, not the time)")
for n, v in zip(names, ring[:5]):
    print(f"  {n:28s} {100.0 * v / tot:6.2f} %")
