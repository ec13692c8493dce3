#!/usr/bin/env python3
"""Frame times of one BASELINE configuration under several option sets (rtr_set_option), same process, alternating runs.
usage (on the GPU box): python scripts/option_bench.py c2 256 "pixel_order=tile" "pixel_order=sorted,sample_chunks=8" ...
Prints, per option set, the median HIP-event frame time over the repeats and the segment count (must be equal everywhere)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import raytracingrust_amd as rt  # noqa: E402
from raytracingrust_amd.configs import baseline_config  # noqa: E402


def main():
    config, spp = sys.argv[1], int(sys.argv[2])
    sets = sys.argv[3:] or [""]
    repeats = int(os.environ.get("REPEATS", "5"))
    size = os.environ.get("SIZE")
    scene, W, H = baseline_config(config, spp)
    if size:
        W, H = (int(v) for v in size.split("x"))
    dev = torch.device("cuda", 0)
    rgba8 = torch.zeros((H, W, 4), dtype=torch.uint8, device=dev)
    ref = None
    renderers = []
    for opts in sets:
        r = rt.Renderer(0)
        for kv in filter(None, opts.split(",")):
            k, v = kv.split("=")
            r.set_option(k, v)
        r.upload_scene(scene)
        renderers.append(r)
    times = [[] for _ in sets]
    for rep in range(repeats + 1):
        for i, r in enumerate(renderers):
            r.render_frame_device(W, H, 1234, rgba8.data_ptr(), 0)
            torch.cuda.synchronize()
            if rep == 0:
                img = rgba8.cpu().numpy().copy()
                if ref is None:
                    ref = img
                same = bool(np.array_equal(img, ref))
                print(f"[{sets[i] or 'default'}] first frame identical to the first set's: {same}; segments {int(r.segment_counts()[-1])}", flush=True)
            else:
                times[i].append(float(r.kernel_times_ms()[-1]))
    for i, opts in enumerate(sets):
        t = np.array(times[i])
        print(f"{opts or 'default':60s} median {np.median(t):8.3f} ms  min {t.min():8.3f}  max {t.max():8.3f}  kernel {renderers[i].kernel_name()[:40]}", flush=True)


if __name__ == "__main__":
    main()
