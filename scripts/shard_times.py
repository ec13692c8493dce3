#!/usr/bin/env python3
"""Per-rank integrate-kernel time of the tile-sharded frame, measured on ONE GPU by launching each rank's share in
turn (what every rank of an N-GPU run executes): shows how the per-GPU kernel time scales with N."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import raytracingrust_amd as rt
from raytracingrust_amd.configs import baseline_config

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
scene, W, H = baseline_config(cfg, spp)
r = rt.Renderer(0)
r.upload_scene(scene)
dev = torch.device("cuda", 0)
buf = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
torch.cuda.synchronize()
base = None
for n in (1, 2, 4, 8):
    from raytracingrust_amd.distributed import sample_range
    for mode in ("tiles", "samples"):
        times = []
        for shard in range(n):
            for seed in (7, 8):
                if mode == "tiles":
                    r.render_tiles_device(W, H, seed, shard, n, buf.data_ptr(), 0)
                else:
                    b, c = sample_range(shard, n, spp)
                    r.render_samples_device(W, H, seed, b, c, buf.data_ptr(), 0)
            times.append(r.stats().kernel_ms)
        worst = max(times)
        base = base or worst
        print(f"{cfg} N={n} {mode:8s}: per-rank kernel ms min {min(times):.3f} max {worst:.3f}  ideal {base / n:.3f}  efficiency {base / n / worst:.3f}")
