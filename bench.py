#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X.

    metric   Mrays/s (ray = one ray segment = one world.hit call of trace_ray, render.rs:38, counted on the
             device) and ms/frame (= ms_per_step)
    workload scenes/benchmark.json at 1920x1080, 256 spp, depth 6, clamp 10 (BASELINE configs[1])
    step     one whole frame: integrate + clamp + mean + sqrt gamma + RGBA8, output resident in HBM

N = 1: one fused kernel per frame.  N > 1 (one process per GPU, launched by torch.distributed.run): the samples
of every pixel are split over the ranks (each rank integrates all pixels for its share of the samples, so every GPU
keeps a full grid of short waves), the per-pixel partial sums are summed onto rank 0 with one RCCL reduce over
xGMI, rank 0 resolves to RGBA8; frame i's reduce overlaps frame i+1's integrate.  `--split tiles` selects the
8x8-tile decomposition instead (bit-identical to one GPU, but under-fills the GPUs).  The total work is the fixed
frame, so scaling is "strong".

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic bytes per ray segment, SURVEY.md section 8(d): 96 (path state in+out) + 20 per sphere tested
# + 32 per node + 48 per triangle (+12 per SkyMap texel); linear world of benchmark.json: 5 spheres, no nodes
BYTES_PER_SEGMENT_C2 = 96 + 20 * 5
# the other configurations: N_node / N_sph / N_tri / sky fraction per segment measured by the oracle under the
# reference's unculled traversal (scripts/bench_configs.py prints them), folded into B_seg
BYTES_PER_SEGMENT = {"c2": 196.0, "c1": 1032.6, "c2b": 317.5, "c3": 1240.7, "c3l": 937.2, "c4": 1480.6, "c5": 1038.4}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s


def committed_pmc(args, world):
    """Counters of the headline kernel from the committed rocprofv3 PMC passes (profiles/r01/traffic.json): they
    cannot be read inside this process, so the figures of the same command's profile run are reported."""
    if world != 1 or args.config or (args.width, args.height, args.spp) != (1920, 1080, 256) or args.bvh:
        return None
    try:
        with open(os.path.join(ROOT, "profiles", "r01", "traffic.json")) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return None


def pmc_traffic(args, world):
    if args.pmc_traffic is not None:
        return args.pmc_traffic
    d = committed_pmc(args, world)
    return float(d["bytes_per_launch"]) if d and "bytes_per_launch" in d else None


def valu_issue(args, world, kernel_ms):
    """The bound that actually binds: VALU issue.  One wave64 VALU instruction occupies a SIMD-32 for >= 2 cycles,
    so the chip issues at most 1024 SIMDs x 2.4e9 / 2 wave-instructions per second (MI355X_MICROARCH.md)."""
    d = committed_pmc(args, world)
    if not d or "sq_insts_valu_per_launch" not in d or kernel_ms <= 0:
        return None
    peak = 1024 * 2.4e9 / 2.0
    ach = float(d["sq_insts_valu_per_launch"]) / (kernel_ms * 1e-3)
    return {"wave_insts_per_launch": float(d["sq_insts_valu_per_launch"]), "achieved_per_s": round(ach, 0),
            "peak_per_s": peak, "frac": round(ach / peak, 4),
            "active_lane_fraction": d.get("active_lane_fraction"),  # SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU * 64)
            "source": "SQ_INSTS_VALU from profiles/r01 (same command), kernel time from this run's HIP events; "
                      "half-rate (compare/select/int-mul) and quarter-rate (sqrt/rcp) instructions make 1.0 unreachable"}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--scene", default=os.path.join(ROOT, "data", "scenes", "benchmark.json"))
    ap.add_argument("--bvh", action="store_true", help="render with enable_bvh_tree=true (not the headline)")
    ap.add_argument("--config", default=None,
                    help="another BASELINE configuration (c1,c2b,c3,c3l,c4,c5: raytracingrust_amd/configs.py) instead "
                         "of the headline scene; width/height come from the configuration")
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5EED5EED5EED5EED)
    ap.add_argument("--cpu-spp", type=int, default=0,
                    help="spp of the bounded CPU-baseline sample (0 = size it for ~15 s of CPU work)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--split", choices=["samples", "tiles"], default="samples",
                    help="multi-GPU decomposition: split the samples of every pixel over the ranks (default: full "
                         "grids of short waves, perfect balance) or 8x8 tiles round-robin (bit-identical to one GPU)")
    ap.add_argument("--mesh-tree", choices=["reference", "sah"], default="reference",
                    help="mesh subtrees: the reference's median split (parity) or the binned-SAH rebuild (faster walk)")
    ap.add_argument("--scene-tree", choices=["reference", "sah"], default="reference",
                    help="scene-level tree over the objects: the reference's median split (parity) or binned SAH")
    ap.add_argument("--pmc-traffic", type=float, default=None,
                    help="HBM bytes per launch from a separate rocprofv3 --pmc pass (profiles/), if collected")
    return ap.parse_args()


def cpu_baseline(scene, width, height, spp_sample, seed):
    """The oracle ("port": C restatement, reference not runnable -- no Rust toolchain) on this host's cores."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import copy

    import pyoracle

    sc = copy.copy(scene)
    sc.settings = copy.copy(scene.settings)
    cores = len(os.sched_getaffinity(0))
    if spp_sample <= 0:
        # size the sample for roughly 15 s of CPU work: probe at 2 spp, then scale (never above the real spp)
        sc.settings.samples_per_pixel = 2
        _, _, pr = pyoracle.OracleScene(sc).render(width, height, seed, threads=cores, want_rgba8=True, want_mean=False)
        spp_sample = int(max(2, min(scene.settings.samples_per_pixel, round(15.0 / max(pr.seconds / 2.0, 1e-6)))))
    sc.settings.samples_per_pixel = spp_sample
    _, _, st = pyoracle.OracleScene(sc).render(width, height, seed, threads=cores, want_rgba8=True, want_mean=False)
    return {
        "value": round(st.segments / st.seconds / 1e6, 3), "unit": "Mrays/s", "cores": int(st.threads), "kind": "port",
        "sample": f"full {width}x{height} frame at {spp_sample} spp (of {scene.settings.samples_per_pixel}), "
                  f"{st.segments} segments in {st.seconds:.2f} s; C restatement of the reference "
                  f"(reference itself not runnable: no Rust toolchain)",
        "ms_per_frame_extrapolated": round(st.seconds * 1e3 * scene.settings.samples_per_pixel / spp_sample, 1),
    }


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    import raytracingrust_amd as rt
    from raytracingrust_amd.distributed import FramePipeline

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # RTR_BENCH_FORCE_DISTRIBUTED=1 (testing aid): take the sharded + RCCL-reduce path even with one rank
    distributed = world > 1 or os.environ.get("RTR_BENCH_FORCE_DISTRIBUTED") == "1"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", device_id=dev, rank=rank, world_size=world)

    if args.config:
        from raytracingrust_amd.configs import baseline_config

        scene, W, H = baseline_config(args.config, args.spp)
        args.scene = args.config
    else:
        scene = rt.Scene.from_file(args.scene)
        scene.settings.samples_per_pixel = args.spp
        if args.bvh:
            scene.settings.enable_bvh_tree = True
        W, H = args.width, args.height

    r = rt.Renderer(local_rank)
    r.set_option("mesh_tree", args.mesh_tree)
    r.set_option("scene_tree", args.scene_tree)
    r.upload_scene(scene)
    rgba8 = torch.zeros((H, W, 4), dtype=torch.uint8, device=dev)
    # N > 1: double-buffered integrate | RCCL reduce + resolve pipeline (raytracingrust_amd/distributed.py)
    pipe = FramePipeline(r, W, H, args.spp, dev, split=args.split) if distributed else None
    # one explicit HIP stream for kernels, torch ops and RCCL alike (a NULL stream would mean "the context's own")
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream

    def step(i):
        seed = args.seed + i
        if not distributed:
            r.render_frame_device(W, H, seed, rgba8.data_ptr(), stream)
        else:
            # zero -> integrate own tiles -> RCCL reduce(SUM) to rank 0 -> resolve on rank 0, frame i's reduce
            # overlapping frame i+1's integrate
            pipe.submit(seed)

    def fence():
        if pipe is not None:
            pipe.wait()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    fence()
    elapsed = time.perf_counter() - t0

    # device-side counters / HIP-event kernel times of the timed launches
    ktimes = r.kernel_times_ms()[-args.steps:]
    segs = r.segment_counts()[-args.steps:]
    my_segments = float(segs.sum())
    my_kernel_ms = float(ktimes.mean()) if len(ktimes) else 0.0
    stats = r.stats()

    t = torch.tensor([elapsed, my_segments, my_kernel_ms], dtype=torch.float64, device=dev)
    if distributed:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        total_segments = float(tsum[1])
        kernel_ms = float(tmax[2])
    else:
        total_segments = my_segments
        kernel_ms = my_kernel_ms

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        mrays = total_segments / elapsed / 1e6
        seg_per_launch = my_segments / max(args.steps, 1)  # rank 0's kernel (N=1: the whole frame)
        pixels_per_launch = stats.primary_samples / max(args.spp, 1)
        b_seg = BYTES_PER_SEGMENT.get(args.config, BYTES_PER_SEGMENT_C2) if args.config else (
            BYTES_PER_SEGMENT["c2b"] if args.bvh else BYTES_PER_SEGMENT_C2)
        alg_bytes = b_seg * seg_per_launch + 4.0 * pixels_per_launch
        achieved = alg_bytes / (my_kernel_ms * 1e-3) / 1e9 if my_kernel_ms > 0 else 0.0
        out = {
            "metric": "Mrays/s", "value": round(mrays, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"{os.path.basename(args.scene)} {W}x{H} {args.spp}spp depth {scene.settings.max_ray_depth} "
                            f"clamp {scene.settings.clamp_indirect:g} "
                            f"{'BVH' if scene.settings.enable_bvh_tree else 'linear'} world",
                "sharding": "single fused kernel" if pipe is None else (
                    f"samples of every pixel split over {world} ranks + RCCL reduce (sum of partial sums)"
                    if pipe.split == "samples" else f"8x8 tiles round-robin over {world} ranks + RCCL reduce"),
                "ms_per_frame": round(ms_per_step, 4),
                "msamples_per_s": round(W * H * args.spp * args.steps / elapsed / 1e6, 2),
                "segments_per_frame": int(total_segments / max(args.steps, 1)),
                "kernel_ms": round(kernel_ms, 4), "vgprs": int(stats.vgprs), "lds_bytes_per_wave": int(stats.lds_bytes),
            },
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(args, world),
                "kernel": "k_render<linear spheres>" if not scene.settings.enable_bvh_tree else "k_render<bvh>",
                "algorithmic_bytes_per_launch": int(alg_bytes), "bytes_per_segment": b_seg,
                "kernel_ms_hip_events": round(my_kernel_ms, 4),
                "valu_issue": valu_issue(args, world, my_kernel_ms),
                "note": "algorithmic bytes are SURVEY 8(d)'s streaming-wavefront model (48 B of path state in and out "
                        "per segment + geometry records); this kernel keeps path state in registers and the scene in "
                        "SGPRs/LDS/L1, so measured HBM traffic is the framebuffer plus the running sums handed between "
                        "sample chunks (~0.17 GB per launch) and frac can exceed 1: the kernel is VALU-issue bound "
                        "(DESIGN.md section 5)",
            },
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(scene, W, H, args.cpu_spp, args.seed)
            out["config"]["gpu_over_cpu"] = round(mrays / max(out["cpu_baseline"]["value"], 1e-9), 1)
        print(json.dumps(out), flush=True)
    r.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
