#!/usr/bin/env python3
"""Diagnostic: lanes active per block of the headline kernel's loop, from a -DRTR_X_LANESTAT build (never the product).
usage (on the GPU box): python scripts/lanestat.py [config] [spp] [WxH] [sample_chunks]"""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RTR_HIP_LIBRARY"] = os.path.join(ROOT, "raytracingrust_amd", "_lib", "librtr_hip_lanestat.so")
import raytracingrust_amd as rt  # noqa: E402
from raytracingrust_amd.configs import baseline_config  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
scene, W, H = baseline_config(cfg, spp)
if len(sys.argv) > 3:
    W, H = (int(v) for v in sys.argv[3].split("x"))
r = rt.Renderer(0)
r.set_option("sample_chunks", sys.argv[4] if len(sys.argv) > 4 else "1")
r.upload_scene(scene)
one = copy.copy(scene)
one.settings = copy.copy(scene.settings)
one.settings.samples_per_pixel = 1
for _ in range(33):
    r.render(8, 8, seed=1, scene=one)  # ring slots the counted launch writes below itself
r.render(W, H, seed=1)
ring = [int(v) for v in r.segment_counts(64)]
c = ring[-2::-1][:32]  # counter-1-k for k = 0..31
seg = ring[-1]
trips = c[0]
print(f"{cfg} {W}x{H}x{spp}: segments {seg}, wave trips {trips}, lanes alive per trip {c[1] / trips:.2f}")
for i in range(5):
    print(f"  sphere {i} root block: executed in {100.0 * c[7 + i] / trips:5.1f} % of trips, {c[2 + i] / max(c[7 + i], 1):5.1f} lanes when executed"
          f" ({c[2 + i] / trips:5.2f} per trip)")
print(f"  hit lanes per trip {c[12] / trips:.2f}; exact-path repeats {c[25]}")
for k, n in ((13, "lambertian"), (14, "metal"), (15, "emission"), (16, "dielectric"), (17, "miss")):
    print(f"  {n:10s} lanes per trip {c[k] / trips:6.2f}")
print(f"  sampler: runs in {100.0 * c[20] / trips:.1f} % of trips, {c[18] / max(c[20], 1):.2f} wave iterations per run, "
      f"{c[19] / max(c[18], 1):.2f} lanes per iteration, {c[19] / trips:.2f} lane-iterations per trip")
print(f"  done: block (chain replay, clamp, accumulate, regenerate) runs in {100.0 * c[22] / trips:.1f} % of trips with "
      f"{c[21] / max(c[22], 1):.2f} lanes")
