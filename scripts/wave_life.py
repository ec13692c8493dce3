#!/usr/bin/env python3
"""Diagnostic: how much of the launch's wave-slot time the headline kernel's workgroups are alive, and how much of their life
passes before the segment loop, from a -DRTR_X_LIFE build (never the product; python -m raytracingrust_amd.build --variant life
-DRTR_X_LIFE).
usage (on the GPU box): python scripts/wave_life.py [config] [spp] [option=value,...]"""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RTR_HIP_LIBRARY"] = os.path.join(ROOT, "raytracingrust_amd", "_lib", "librtr_hip_life.so")
import raytracingrust_amd as rt  # noqa: E402
from raytracingrust_amd.configs import baseline_config  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
opts = sys.argv[3] if len(sys.argv) > 3 else ""
scene, W, H = baseline_config(cfg, spp)
r = rt.Renderer(0)
for kv in filter(None, opts.split(",")):
    k, v = kv.split("=")
    r.set_option(k, v)
r.upload_scene(scene)
one = copy.copy(scene)
one.settings = copy.copy(scene.settings)
one.settings.samples_per_pixel = 1
SLOTS = 8192  # 256 CUs x 4 SIMDs x 8 waves
for rep in range(3):
    for _ in range(3):
        r.render(8, 8, seed=1, scene=one)  # ring slots the stamped launch writes below itself
    r.render(W, H, seed=1)
    ring = [int(v) for v in r.segment_counts(8)]
    life, before, wgs = ring[-2], ring[-3], ring[-4]
    ms = float(r.kernel_times_ms()[-1])
    slot_ticks = SLOTS * ms * 1e5  # s_memrealtime: 100 MHz
    print(f"{cfg} {W}x{H}x{spp} [{opts or 'default'}]: kernel {ms:.3f} ms, {wgs} workgroups; alive {100.0 * life / slot_ticks:.1f} % of "
          f"{SLOTS} slots x launch; before the loop {100.0 * before / life:.2f} % of life ({before / wgs / 100.0:.2f} us per workgroup); "
          f"mean life {life / wgs / 100.0:.1f} us", flush=True)
