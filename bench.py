#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X.

    metric   Mrays/s (ray = one ray segment = one world.hit call of trace_ray, render.rs:38, counted on the
             device) and ms/frame (= ms_per_step)
    workload scenes/benchmark.json at 1920x1080, 256 spp, depth 6, clamp 10 (BASELINE configs[1])
    step     one whole frame: integrate + clamp + mean + sqrt gamma + RGBA8, output resident in HBM

N = 1: one fused kernel per frame.  N > 1: one process per GPU.  `python bench.py --gpus N` launches the N ranks
itself (python -m torch.distributed.run, rendezvous on 127.0.0.1) before anything touches a GPU; started by a
launcher (WORLD_SIZE set) it is one of the ranks.  The samples of every pixel are split over the ranks (each rank
integrates all pixels for its share of the samples, so every GPU keeps a full grid of short waves), the per-pixel
partial sums are summed onto rank 0 with one RCCL reduce over xGMI, rank 0 resolves to RGBA8; frame i's reduce
overlaps frame i+1's integrate.  `--split tiles` selects the 8x8-tile decomposition instead (bit-identical to one
GPU, but under-fills the GPUs).  The total work is the fixed frame, so scaling is "strong".

Prints ONE JSON line on rank 0.  The `roofline` object names the bound that binds this path -- VALU instruction
issue (there is no dense contraction and the scene is cache resident: DESIGN.md section 5) -- from counters of a
rocprofv3 run of the same command on the same kernel build (profiles/, stamped with the kernel sources' hash; null
when the build has changed since), with the measured HBM traffic beside it.  At N = 1 the CPU leg renders the same
seed as one GPU frame and the two are compared byte for byte (`config.parity_full_frame`).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic bytes per ray segment, SURVEY.md section 8(d): 96 (path state in+out) + 20 per sphere tested
# + 32 per node + 48 per triangle (+12 per SkyMap texel); N_node / N_sph / N_tri / sky fraction per segment measured
# by the oracle under the reference's unculled traversal (scripts/bench_configs.py prints them).  A streaming-wavefront
# MODEL: this design keeps path state in registers, so the figure is reported as an auxiliary, never as the roofline.
BYTES_PER_SEGMENT = {"c2": 196.0, "c1": 1032.6, "c2b": 317.5, "c3": 1240.7, "c3l": 937.2, "c4": 1480.6, "c5": 1038.4}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
# One wave64 VALU instruction occupies a SIMD-32 for 2 cycles: 256 CUs x 4 SIMDs x 2.4 GHz / 2 (MI355X_MICROARCH.md,
# "Wave scheduling" and the cycle-constants table).  Half-rate (compare / select / integer multiply) and quarter-rate
# (sqrt / rcp) instructions and the clock the chip holds under load (< 2.4 GHz) keep 1.0 out of reach.
VALU_PEAK_WAVE_INSTS_PER_S = 256 * 4 * 2.4e9 / 2.0
PROFILE_ROUND = "r03"
# The tree kernels' bound (round 3): the vector L1's gather rate.  Every node step gathers a per-lane 56-byte record (three
# 16-byte loads + one 8-byte load), every leaf test 36 (triangle) or 16 (sphere: centre + r^2) bytes, and a wave's gather of per-lane
# 64-byte-aligned records is served at 18-21 B per clock and CU whether the table sits in L1 or L2 -- measured with
# scripts/ubench/gather.hip (profiles/r02/evidence/ubench_gather.txt: 56 B per lane, 16 KB table 12 795 GB/s, 1 MB table
# 11 965 GB/s chip-wide).  Peak = the mean of the two; achieved = bytes the walk gathered (device-counted node steps and leaf
# tests of one frame, option "traversal_counters") over the integrate kernel's time.
L1_GATHER_PEAK_GBS = 12380.0
NODE_GATHER_BYTES, TRIANGLE_GATHER_BYTES, SPHERE_GATHER_BYTES = 56.0, 36.0, 16.0


def parse_args(argv=None):
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
    ap.add_argument("--mesh-tree", choices=["reference", "sah", "lbvh"], default="reference",
                    help="mesh subtrees: the reference's median split (parity) or the binned-SAH rebuild (faster walk)")
    ap.add_argument("--scene-tree", choices=["reference", "sah"], default="reference",
                    help="scene-level tree over the objects: the reference's median split (parity) or binned SAH")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="headline run only: skip the short secondary measurements of BASELINE's tree configurations")
    ap.add_argument("--native", action="store_true",
                    help="N GPUs from ONE process through the C ABI's multi-GPU context (rtr_create_multi: a host thread per "
                         "GPU, sample split, ncclReduce and resolve inside librtr_hip.so) instead of one torch.distributed "
                         "rank per GPU; same JSON line")
    ap.add_argument("--no-traversal-counters", action="store_true",
                    help="tree worlds: skip the extra counting frame behind roofline.achieved (profiling runs: keeps the dispatch list clean)")
    ap.add_argument("--tree-kernel", default=None, help="rtr_set_option tree_kernel (auto, stream, samples, resumable, wave)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------
# N > 1 without a launcher: start the ranks
# ---------------------------------------------------------------------------------------------------------------
def plumbing_mode() -> bool:
    """RTR_BENCH_PLUMBING=1 (tests/test_bench_contract.py): the rank processes run on the CPU over gloo with a
    stand-in integrator, so that launching, rendezvous, sample ranges and the reduce can be exercised without GPUs."""
    return os.environ.get("RTR_BENCH_PLUMBING") == "1"


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def visible_gpus() -> int:
    """GPU agents of the KFD topology (nodes with a non-zero simd_count), capped by the *_VISIBLE_DEVICES lists -- read
    from sysfs, so the launcher parent never initialises a GPU (a torch call here could, depending on the build)."""
    n = 0
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        for node in os.listdir(base):
            try:
                with open(os.path.join(base, node, "properties")) as fh:
                    props = dict(line.split()[:2] for line in fh if len(line.split()) >= 2)
                if int(props.get("simd_count", "0")) > 0:
                    n += 1
            except (OSError, ValueError):
                continue
    except OSError:
        return 0
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def spawn_ranks(args) -> int:
    """Called before any HIP / torch.cuda use in this process: starts args.gpus fresh rank processes (never re-execs a
    process that has touched the GPU) and returns their exit code.  Rank 0 prints the JSON line on the shared stdout."""
    if not plumbing_mode():
        have = visible_gpus()  # from sysfs: nothing in the launcher process opens KFD / HSA
        if have < args.gpus:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible; refusing to run on fewer\n")
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def plumbing_rank(args, world, rank) -> int:
    """The CPU stand-in of a rank (see plumbing_mode): same launch, rendezvous, sample split, reduce and JSON contract."""
    import torch
    import torch.distributed as dist

    from raytracingrust_amd.distributed import sample_range

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    begin, count = sample_range(rank, world, args.spp)
    w, h = 16, 8
    # stand-in integrate: sample s of pixel p contributes (p + 1) * (s + 1) -- an exact small-integer sum in f32
    pix = torch.arange(1, w * h + 1, dtype=torch.float32).reshape(h, w, 1).repeat(1, 1, 3)
    part = pix * float(sum(s + 1 for s in range(begin, begin + count)))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        acc = part.clone()
        dist.reduce(acc, dst=0, op=dist.ReduceOp.SUM)
    dist.barrier()
    elapsed = time.perf_counter() - t0
    ranges = [None] * world
    dist.all_gather_object(ranges, (begin, count))
    if rank == 0:
        want = pix * float(args.spp * (args.spp + 1) // 2)
        print(json.dumps({"metric": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "plumbing": True, "sample_ranges": ranges, "reduce_matches_single_rank": bool(torch.equal(acc, want)),
                          "ms_per_step": round(elapsed * 1e3 / max(args.steps, 1), 4)}), flush=True)
    dist.destroy_process_group()
    return 0


# ---------------------------------------------------------------------------------------------------------------
# counters of the rocprofv3 run (profiles/), stamped with the hash of the kernel sources they were taken on
# ---------------------------------------------------------------------------------------------------------------
def workload_key(args, W, H):
    name = args.config or ("c2b" if args.bvh else "c2")
    if not args.config and os.path.basename(args.scene) != "benchmark.json":
        name = os.path.splitext(os.path.basename(args.scene))[0]
    key = f"{name}_{W}x{H}_{args.spp}spp"
    if args.mesh_tree == "sah" or args.scene_tree == "sah":
        key += "_sah"
    if args.mesh_tree == "lbvh":
        key += "_lbvh"
    if args.tree_kernel:
        key += "_" + args.tree_kernel
    return key


def committed_profile(args, W, H, world):
    """(dict | None, status): the PMC summary scripts/profile_bench.sh wrote for this workload, if it was taken on
    the kernel sources this process runs."""
    if world != 1:
        return None, "single-GPU profile only"
    from raytracingrust_amd.build import kernel_sha

    path = os.path.join(ROOT, "profiles", PROFILE_ROUND, workload_key(args, W, H) + ".json")
    try:
        with open(path) as fh:
            d = json.load(fh)
    except (OSError, ValueError):
        return None, "no profile for this workload under profiles/" + PROFILE_ROUND
    if d.get("kernel_sha") != kernel_sha():
        return None, f"stale: profiled at kernel_sha {d.get('kernel_sha')}, running {kernel_sha()}"
    return d, "profiles/%s/%s.json" % (PROFILE_ROUND, workload_key(args, W, H))


def roofline_block(prof, prof_status, integrate_ms, alg_bytes, kernel_name, walk=None):
    """Flat scalars only.  Sphere world: bound = VALU issue; tree worlds (walk given): bound = the vector L1's gather rate,
    the VALU figures beside it.  The HBM side is what the counters measured, not a model."""
    out = {"bound": "valu_issue", "achieved": None, "peak": round(VALU_PEAK_WAVE_INSTS_PER_S / 1e9, 1),
           "unit": "G wave-instructions/s", "frac": None, "traffic": None,
           "active_lane_fraction": None, "lane_throughput_frac": None,
           "hbm_achieved_gbs": None, "hbm_peak_gbs": HBM_PEAK_GBS, "hbm_frac_measured": None,
           "kernel": kernel_name, "kernel_ms_hip_events": round(integrate_ms, 4),
           "counters": prof_status,
           "aux_streaming_model_gbs": round(alg_bytes / (integrate_ms * 1e-3) / 1e9, 1) if integrate_ms > 0 else None,
           "aux_streaming_model_note": "SURVEY 8(d)'s algorithmic bytes (48 B of path state in and out per segment + "
                                       "geometry records) over the kernel time: a model of a streaming design, bytes "
                                       "this register-resident kernel never moves; not a bound"}
    if walk is not None and integrate_ms > 0:
        # tree world: what binds is the vector L1 (see L1_GATHER_PEAK_GBS); the VALU-issue figures move to valu_* keys
        gbs = walk["gather_bytes"] / (integrate_ms * 1e-3) / 1e9
        out.update({"bound": "vector_l1_gather", "achieved": round(gbs, 1), "peak": L1_GATHER_PEAK_GBS, "unit": "GB/s",
                    "frac": round(gbs / L1_GATHER_PEAK_GBS, 4),
                    "gather_bytes_per_launch": walk["gather_bytes"], "node_steps_per_segment": walk["node_steps_per_segment"],
                    "leaf_tests_per_segment": walk["leaf_tests_per_segment"],
                    "sky_lookups_per_frame": walk["sky_lookups_per_frame"],
                    "peak_source": "scripts/ubench/gather.hip on this GPU type: per-lane 56-byte record gathers, L1- and L2-resident "
                                   "tables (profiles/r02/evidence/ubench_gather.txt)"})
    if prof is None or integrate_ms <= 0:
        return out
    t = integrate_ms * 1e-3
    ach = float(prof["sq_insts_valu"]) / t
    if walk is not None:
        out["valu_achieved_g_wave_insts_per_s"] = round(ach / 1e9, 1)
        out["valu_issue_frac"] = round(ach / VALU_PEAK_WAVE_INSTS_PER_S, 4)
    else:
        out["achieved"] = round(ach / 1e9, 1)
        out["frac"] = round(ach / VALU_PEAK_WAVE_INSTS_PER_S, 4)
    if prof.get("active_lane_fraction") is not None:
        out["active_lane_fraction"] = round(float(prof["active_lane_fraction"]), 4)
        out["lane_throughput_frac"] = round(ach / VALU_PEAK_WAVE_INSTS_PER_S * out["active_lane_fraction"], 4)
    if prof.get("hbm_bytes_per_launch") is not None:
        out["traffic"] = float(prof["hbm_bytes_per_launch"])
        out["hbm_achieved_gbs"] = round(out["traffic"] / t / 1e9, 2)
        out["hbm_frac_measured"] = round(out["traffic"] / t / 1e9 / HBM_PEAK_GBS, 5)
    out["kernel_avg_ms_rocprof"] = round(float(prof.get("kernel_avg_ns_rocprof", 0.0)) / 1e6, 4)
    out["vgprs_profiled"] = prof.get("compiled_vgprs", prof.get("vgpr"))
    out["sgprs_profiled"] = prof.get("compiled_sgprs", prof.get("sgpr"))
    out["sgpr_spills_profiled"] = prof.get("compiled_sgpr_spills")
    out["vgpr_spills_profiled"] = prof.get("compiled_vgpr_spills")
    return out


# ---------------------------------------------------------------------------------------------------------------
# CPU leg: the oracle on this host's cores, and the byte-for-byte check of one GPU frame against it
# ---------------------------------------------------------------------------------------------------------------
def cpu_baseline(scene, width, height, spp_sample, seed):
    """The oracle ("port": C restatement, reference not runnable -- no Rust toolchain) on this host's cores.
    Returns (record, rgba8 of the rendered sample, its spp, its segment count)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import copy

    import pyoracle

    sc = copy.copy(scene)
    sc.settings = copy.copy(scene.settings)
    cores = len(os.sched_getaffinity(0))
    if spp_sample <= 0:
        # size the sample for roughly 15 s of CPU work: probe at 2 spp, then scale (never above the real spp)
        sc.settings.samples_per_pixel = 2
        _, _, pr = pyoracle.OracleScene(sc, trig_mode=pyoracle.TRIG_POLY).render(width, height, seed, threads=cores, want_rgba8=True, want_mean=False)
        spp_sample = int(max(2, min(scene.settings.samples_per_pixel, round(15.0 / max(pr.seconds / 2.0, 1e-6)))))
    sc.settings.samples_per_pixel = spp_sample
    rgba, _, st = pyoracle.OracleScene(sc, trig_mode=pyoracle.TRIG_POLY).render(width, height, seed, threads=cores, want_rgba8=True, want_mean=False)
    rec = {
        "unculled_box_tests_per_segment": round(st.box_tests / max(st.segments, 1), 2),
        "unculled_leaf_tests_per_segment": round((st.sphere_tests + st.tri_tests) / max(st.segments, 1), 2),
        "value": round(st.segments / st.seconds / 1e6, 3), "unit": "Mrays/s", "cores": int(st.threads), "kind": "port",
        "sample": f"full {width}x{height} frame at {spp_sample} spp (of {scene.settings.samples_per_pixel}), "
                  f"{st.segments} segments in {st.seconds:.2f} s; C restatement of the reference "
                  f"(reference itself not runnable: no Rust toolchain; parity unpinned: the reference holds no fixtures)",
        "ms_per_frame_extrapolated": round(st.seconds * 1e3 * scene.settings.samples_per_pixel / spp_sample, 1),
    }
    return rec, rgba, spp_sample, int(st.segments)


def other_configs(r, torch, dev, stream, seed):
    """Short secondary measurements, reported beside the headline so that the driver's record carries them too:
    BASELINE's tree-world configurations (C1 simple.json, C3 glass monkey + HDRI, C4 house + plane + HDRI at 1920x1080, C5
    simple.json + HDRI at 3840x2160).  Per configuration: (a) a full-size frame at 4 spp rendered on the GPU and by the
    oracle on the same seed, compared byte for byte and segment for segment -> `cN_parity`; (b) one warm-up frame and two
    timed frames at 256 spp through the device-resident entry point; (c) for C3 and C5 one timed frame at the spp BASELINE
    states (512 / 1024).  Flat scalars."""
    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    from raytracingrust_amd.configs import baseline_config

    cores = len(os.sched_getaffinity(0))
    out = {}

    def timed(W, H, rgba8, frames, first_seed, scene=None):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(frames):
            r.render_frame_device(W, H, first_seed + i, rgba8.data_ptr(), stream, scene=scene)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / frames
        return dt, float(r.segment_counts()[-frames:].mean())

    for name, stated_spp in (("c1", 256), ("c3", 512), ("c4", 256), ("c5", 1024)):
        scene, W, H = baseline_config(name, 256)
        r.upload_scene(scene)
        rgba8 = torch.zeros((H, W, 4), dtype=torch.uint8, device=dev)
        # (a) parity at full size, low spp: the checker renders the same seed (polynomial trig: the kernels' own)
        low = copy_settings(scene, 4)
        r.render_frame_device(W, H, seed, rgba8.data_ptr(), stream, scene=low)
        torch.cuda.synchronize()
        g_seg, g_rgba = int(r.segment_counts()[-1]), rgba8.cpu().numpy()
        o_rgba, _, o_st = pyoracle.OracleScene(low, trig_mode=pyoracle.TRIG_POLY).render(W, H, seed, threads=cores, want_rgba8=True, want_mean=False)
        bad = int(np.count_nonzero(np.any(g_rgba != o_rgba, axis=-1)))
        out[name + "_parity"] = "bit-exact" if bad == 0 and g_seg == int(o_st.segments) else "MISMATCH"
        out[name + "_parity_detail"] = f"{W}x{H} at 4 spp: {bad} RGBA8 pixels differ; segments GPU {g_seg} vs oracle {int(o_st.segments)}"
        # (b) 256 spp
        r.render_frame_device(W, H, seed, rgba8.data_ptr(), stream)
        dt, seg = timed(W, H, rgba8, 2, seed + 1)
        out[name + "_mrays_per_s"] = round(seg / dt / 1e6, 1)
        out[name + "_ms_per_frame"] = round(dt * 1e3, 3)
        out[name + "_segments_per_frame"] = int(seg)
        # (c) BASELINE's stated spp
        if stated_spp != 256:
            full = copy_settings(scene, stated_spp)
            dt, seg = timed(W, H, rgba8, 1, seed + 7, scene=full)
            out[f"{name}_{stated_spp}spp_mrays_per_s"] = round(seg / dt / 1e6, 1)
            out[f"{name}_{stated_spp}spp_ms_per_frame"] = round(dt * 1e3, 3)
        del rgba8
    out["note"] = ("reference (median-split) trees, tree_kernel auto; 1920x1080: C1 simple.json 102 spheres, C3 glass monkey.obj + "
                   "plane + synthetic HDRI, C4 house.obj + plane + synthetic HDRI depth 3; 3840x2160: C5 simple.json + HDRI; cN_* at 256 "
                   "spp (two timed frames), cN_<spp>spp_* one timed frame at BASELINE's stated spp, cN_parity: a full-size 4-spp frame "
                   "against the oracle on the same seed in this run; full lines: python bench.py --config cN")
    return out


def copy_settings(scene, spp):
    import copy

    sc = copy.copy(scene)
    sc.settings = copy.copy(scene.settings)
    sc.settings.samples_per_pixel = spp
    return sc


def main():
    args = parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.native:
        if int(os.environ.get("RANK", "0")) != 0:
            return  # started under a launcher: the native path is one process, rank 0's
        if args.gpus > 1 and visible_gpus() < args.gpus:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} but only {visible_gpus()} GPU(s) are visible; refusing to run on fewer\n")
            sys.exit(2)
        world_env = "1"
    if world_env is None and args.gpus > 1:
        sys.exit(spawn_ranks(args))  # nothing in this process has touched a GPU
    world = int(world_env or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and not args.native:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU (or let bench.py start them: "
                         f"run it without a launcher)")
    if plumbing_mode():
        sys.exit(plumbing_rank(args, world, rank))

    import numpy as np
    import torch
    import torch.distributed as dist

    import raytracingrust_amd as rt
    from raytracingrust_amd.distributed import FramePipeline

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
        scene_name = args.config
    else:
        scene = rt.Scene.from_file(args.scene)
        scene.settings.samples_per_pixel = args.spp
        if args.bvh:
            scene.settings.enable_bvh_tree = True
        W, H = args.width, args.height
        scene_name = os.path.basename(args.scene)

    # --native: one context over GPUs 0 .. N-1 (N = 1 included: the same threads, reduce and resolve with one rank)
    r = rt.Renderer(list(range(args.gpus))) if args.native else rt.Renderer(local_rank)
    if args.native and args.split == "tiles":
        r.set_option("multi_split", "tiles")
    r.set_option("mesh_tree", args.mesh_tree)
    r.set_option("scene_tree", args.scene_tree)
    if args.tree_kernel:
        r.set_option("tree_kernel", args.tree_kernel)
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
            # integrate own samples (or tiles) -> RCCL reduce(SUM) to rank 0 -> resolve on rank 0, frame i's reduce
            # overlapping frame i+1's integrate
            pipe.submit(seed)

    def fence():
        if pipe is not None:
            pipe.wait()
        if args.native:
            r.stats()  # waits for every GPU of the context (and raises on a device fault)
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

    # device-side counters / HIP-event kernel times of the timed launches (rtr_* raise if a launch set the fault word)
    ktimes = r.kernel_times_ms()[-args.steps:]
    itimes = r.integrate_times_ms()[-args.steps:]
    segs = r.segment_counts()[-args.steps:]
    my_segments = float(segs.sum())
    my_kernel_ms = float(ktimes.mean()) if len(ktimes) else 0.0
    my_integrate_ms = float(itimes.mean()) if len(itimes) else 0.0
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
        pixels_per_launch = stats.primary_samples / max(args.spp // world if (distributed and pipe.split == "samples") else args.spp, 1)
        b_seg = BYTES_PER_SEGMENT.get(args.config, BYTES_PER_SEGMENT["c2"]) if args.config else (
            BYTES_PER_SEGMENT["c2b"] if args.bvh else BYTES_PER_SEGMENT["c2"])
        alg_bytes = b_seg * seg_per_launch + 4.0 * pixels_per_launch
        prof, prof_status = committed_profile(args, W, H, args.gpus if args.native else world)
        walk = None
        timed_kernel = r.kernel_name()
        if scene.settings.enable_bvh_tree and world == 1 and not args.native and not args.tree_kernel and not args.no_traversal_counters:
            # one more frame of the last timed seed through the counting instantiation of the samples kernel: what the walk gathered
            # (the same visits whichever tree kernel the timed frames took: every lane walks its ray the same way in all of them)
            r.set_option("traversal_counters", "on")
            r.render_frame_device(W, H, args.seed + args.warmup + args.steps - 1, rgba8.data_ptr(), stream)
            cst = r.stats()
            r.set_option("traversal_counters", "off")
            if cst.traversal_counted:
                info = r.scene_info()
                tri_share = info.n_triangles / max(info.n_leaves, 1)
                leaf_bytes = tri_share * TRIANGLE_GATHER_BYTES + (1.0 - tri_share) * SPHERE_GATHER_BYTES
                walk = {"gather_bytes": float(cst.node_steps) * NODE_GATHER_BYTES + float(cst.leaf_tests) * leaf_bytes,
                        "node_steps_per_segment": round(cst.node_steps / max(cst.segments, 1), 2),
                        "leaf_tests_per_segment": round(cst.leaf_tests / max(cst.segments, 1), 2),
                        "sky_lookups_per_frame": int(cst.sky_lookups)}
        out = {
            "metric": "Mrays/s", "value": round(mrays, 2), "unit": "Mrays/s", "n_gpus": args.gpus if args.native else world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"{scene_name} {W}x{H} {args.spp}spp depth {scene.settings.max_ray_depth} "
                            f"clamp {scene.settings.clamp_indirect:g} "
                            f"{'BVH' if scene.settings.enable_bvh_tree else 'linear'} world",
                "sharding": (f"native: one process, rtr_create_multi over {args.gpus} GPU(s), {args.split} split + ncclReduce + resolve "
                             f"inside librtr_hip.so") if args.native else "single fused kernel" if pipe is None else (
                    f"samples of every pixel split over {world} ranks + RCCL reduce (sum of partial sums)"
                    if pipe.split == "samples" else f"8x8 tiles round-robin over {world} ranks + RCCL reduce"),
                "mesh_tree": args.mesh_tree, "scene_tree": args.scene_tree,
                "ms_per_frame": round(ms_per_step, 4),
                "msamples_per_s": round(W * H * args.spp * args.steps / elapsed / 1e6, 2),
                "segments_per_frame": int(total_segments / max(args.steps, 1)),
                "kernel_ms": round(kernel_ms, 4), "vgprs": int(stats.vgprs), "lds_bytes_per_wave": int(stats.lds_bytes),
                "upload_ms": round(float(stats.upload_ms), 3),
                "parity": "unpinned (the reference holds no fixtures and cannot be built here); checked against oracle/",
            },
            "roofline": roofline_block(prof, prof_status, my_integrate_ms, alg_bytes, timed_kernel, walk),
        }
        if not args.no_cpu_baseline and world == 1 and not (args.native and args.gpus > 1):
            rec, o_rgba, o_spp, o_segments = cpu_baseline(scene, W, H, args.cpu_spp, args.seed)
            out["cpu_baseline"] = rec
            out["config"]["gpu_over_cpu"] = round(mrays / max(rec["value"], 1e-9), 1)
            # the same seed, size and spp on the GPU, compared byte for byte and segment for segment
            import copy

            sc = copy.copy(scene)
            sc.settings = copy.copy(scene.settings)
            sc.settings.samples_per_pixel = o_spp
            r.render_frame_device(W, H, args.seed, rgba8.data_ptr(), stream, scene=sc)
            torch.cuda.synchronize()
            g_segments = int(r.segment_counts()[-1])
            g_rgba = rgba8.cpu().numpy()
            bad_px = int(np.count_nonzero(np.any(g_rgba != o_rgba, axis=-1)))
            max_diff = int(np.max(np.abs(g_rgba.astype(np.int16) - o_rgba.astype(np.int16)))) if bad_px else 0
            out["config"]["parity_full_frame"] = ("bit-exact" if bad_px == 0 and g_segments == o_segments else "MISMATCH")
            out["config"]["parity_full_frame_detail"] = (
                f"{W}x{H} at {o_spp} spp, seed {args.seed:#x}: {bad_px} of {W * H} RGBA8 pixels differ (max channel "
                f"difference {max_diff}); segments GPU {g_segments} vs oracle {o_segments}")
        headline = not args.config and not args.bvh and (W, H, args.spp) == (1920, 1080, 256) and os.path.basename(args.scene) == "benchmark.json"
        if headline and world == 1 and not args.no_other_configs:
            out["other_configs"] = other_configs(r, torch, dev, stream, args.seed)
        print(json.dumps(out), flush=True)
    r.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
