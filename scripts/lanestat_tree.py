#!/usr/bin/env python3
"""Diagnostic: lanes active per block of the sample-parallel tree kernel's loop (k_render_bvh_sp), from a
-DRTR_X_LANESTAT build (never the product).  usage (on the GPU box): python scripts/lanestat_tree.py [config] [spp] [WxH]"""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RTR_HIP_LIBRARY"] = os.path.join(ROOT, "raytracingrust_amd", "_lib", "librtr_hip_lanestat.so")
import raytracingrust_amd as rt  # noqa: E402
from raytracingrust_amd.configs import baseline_config  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
scene, W, H = baseline_config(cfg, spp)
if len(sys.argv) > 3:
    W, H = (int(v) for v in sys.argv[3].split("x"))
r = rt.Renderer(0)
r.set_option("tree_kernel", "samples")
r.upload_scene(scene)
one = copy.copy(scene)
one.settings = copy.copy(scene.settings)
one.settings.samples_per_pixel = 1
for _ in range(33):
    r.render(8, 8, seed=1, scene=one)  # ring slots the counted launch writes below itself
r.render(W, H, seed=1)
ring = [int(v) for v in r.segment_counts(64)]
c = ring[-2::-1][:32]
seg = ring[-1]
print(f"{cfg} {W}x{H}x{spp}: segments {seg}; outer iterations {c[0]} with {c[15] / max(c[0], 1):.1f} lanes alive")
print(f"  walk trips {c[1]} ({c[1] / seg * 64:.1f} per 64 segments), {c[2] / max(c[1], 1):.1f} lanes walking per trip; "
      f"lane-trips per segment {c[2] / seg:.2f}")
print(f"  leaf block: in {100.0 * c[4] / c[1]:.1f} % of trips with {c[3] / max(c[4], 1):.1f} lanes ({c[3] / seg:.2f} leaf tests per segment)")
print(f"  node block: in {100.0 * c[6] / c[1]:.1f} % of trips with {c[5] / max(c[6], 1):.1f} lanes ({c[5] / seg:.2f} node steps per segment); "
      f"both children entered: {c[14] / max(c[5], 1) * 100:.1f} % of node steps")
print(f"  pop block: in {100.0 * c[8] / c[1]:.1f} % of trips with {c[7] / max(c[8], 1):.1f} lanes; loop {c[9] / max(c[8], 1):.2f} wave iterations "
      f"per run, {c[10] / max(c[9], 1):.1f} lanes per iteration")
print(f"  shading: {c[11]} runs ({c[11] / max(c[0], 1):.2f} per outer iteration) with {c[12] / max(c[11], 1):.1f} lanes")
