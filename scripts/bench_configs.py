#!/usr/bin/env python3
"""Secondary benchmarks: BASELINE.json's other single-GPU configurations (not the bench.py contract line).

  c1  scenes/simple.json (102 spheres, BVH world), 1920x1080
  c2b scenes/benchmark.json with enable_bvh_tree=true
  c3  monkey.obj + plane + synthetic 2048x1024 HDRI, camera (0,0,3)->(0,0,0) fov 60, 16:9
  c4  src/gui.rs:56-119 scene: house.obj + plane.obj (mix ground), synthetic HDRI, depth 3
  c5  simple.json geometry + HDRI at 3840x2160
Prints one JSON line per config: Mrays/s, ms/frame, kernel ms and the algorithmic-bytes roofline figure with
N_node / N_sph / N_tri per segment measured by the oracle (reference traversal semantics) on a small frame."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402

import raytracingrust_amd as rt  # noqa: E402
from raytracingrust_amd.configs import baseline_config  # noqa: E402


def build(name, spp):
    return baseline_config(name, spp)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="c1,c2b,c3,c3l,c4")
    ap.add_argument("--spp", type=int, default=32)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--oracle-size", default="96x54")
    ap.add_argument("--options", default="", help="comma-separated name=value pairs for rtr_set_option (e.g. mesh_tree=sah)")
    args = ap.parse_args()
    import pyoracle

    ow, oh = (int(v) for v in args.oracle_size.split("x"))
    r = rt.Renderer(0)
    for kv in filter(None, args.options.split(",")):
        r.set_option(*kv.split("=", 1))
    for name in args.configs.split(","):
        scene, W, H = build(name, args.spp)
        t0 = time.perf_counter()
        r.upload_scene(scene)
        up_ms = (time.perf_counter() - t0) * 1e3
        r.render(W, H, seed=1)  # warm-up
        t0 = time.perf_counter()
        for i in range(args.steps):
            r.render(W, H, seed=2 + i)
        wall = (time.perf_counter() - t0) / args.steps
        st = r.stats()
        info = r.scene_info()
        # reference-traversal counts per segment from the oracle on a small frame
        small = build(name, 2)[0]
        _, _, ost = pyoracle.OracleScene(small, trig_mode=pyoracle.TRIG_POLY).render(ow, oh, 1, threads=0, want_mean=False)
        n_node, n_sph, n_tri = (ost.box_tests / ost.segments, ost.sphere_tests / ost.segments, ost.tri_tests / ost.segments)
        sky_frac = ost.sky_lookups / ost.segments
        b_seg = 96 + 32 * n_node + 20 * n_sph + 48 * n_tri + 12 * sky_frac
        alg = b_seg * st.segments + 4.0 * W * H
        print(json.dumps({
            "config": name, "size": f"{W}x{H}", "spp": args.spp, "mrays_per_s": round(st.segments / (st.kernel_ms * 1e-3) / 1e6, 1),
            "ms_per_frame_wall_incl_d2h": round(wall * 1e3, 3), "kernel_ms": round(st.kernel_ms, 3),
            "segments": int(st.segments), "seg_per_sample": round(st.segments / (W * H * args.spp), 3),
            "upload_ms": round(up_ms, 1), "nodes": info.n_nodes, "leaves": info.n_leaves, "height": info.tree_height,
            "vgprs": st.vgprs, "lds_bytes": st.lds_bytes,
            "oracle_per_segment": {"N_node": round(n_node, 2), "N_sph": round(n_sph, 2), "N_tri": round(n_tri, 2), "sky": round(sky_frac, 3)},
            "bytes_per_segment": round(b_seg, 1), "achieved_GBs": round(alg / (st.kernel_ms * 1e-3) / 1e9, 1),
            "frac_of_8TBs": round(alg / (st.kernel_ms * 1e-3) / 8e12, 4),
        }), flush=True)
    r.close()


if __name__ == "__main__":
    main()
