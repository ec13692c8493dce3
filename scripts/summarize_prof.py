#!/usr/bin/env python3
"""Condense rocprofv3 csv output (kernel stats + PMC counters) of scripts/profile_bench.sh into text tables and into
the JSON record bench.py reads (profiles/<round>/<workload>.json): per launch of the dominant integrate kernel, stamped
with the hash of the kernel sources.  usage: summarize_prof.py <prof dir> [the bench.py arguments that were profiled]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = sys.argv[1]
bench_args = sys.argv[2:]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


def is_integrate(name):
    return "k_render" in name or "k_wf_" in name


print("== kernel stats ==")
stats = {}
for f in find("trace/**/*kernel_stats.csv"):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            print({k: row[k] for k in row if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})
            stats[row["Name"]] = row
print("== kernel trace (integrate-kernel dispatches) ==")
regs = {}
durations = defaultdict(list)
for f in find("trace/**/*kernel_trace.csv"):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")
            if is_integrate(name) or "k_sum_samples" in name:
                d = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
                durations[name].append(d)
                regs[name] = {"vgpr": row.get("VGPR_Count"), "accum_vgpr": row.get("Accum_VGPR_Count"), "sgpr": row.get("SGPR_Count"),
                              "lds": row.get("LDS_Block_Size"), "scratch": row.get("Scratch_Size")}
                print(name[:70], "dur_ns", d, regs[name], "grid", row.get("Grid_Size"), "wg", row.get("Workgroup_Size"))
# the dominant kernel: largest total duration among the integrate kernels
dominant = max((n for n in durations if is_integrate(n)), key=lambda n: sum(durations[n]), default=None)
print("== PMC (per dispatch of each integrate kernel: mean over dispatches) ==")
pmc = defaultdict(dict)
totals = defaultdict(dict)
for d in ("pmc_sq1", "pmc_sq2", "pmc_fetch", "pmc_write"):
    acc = defaultdict(list)
    for f in find(d + "/**/*counter_collection.csv"):
        with open(f) as fh:
            per = defaultdict(float)
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "")
                if not (is_integrate(name) or "k_sum_samples" in name):
                    continue
                per[(name, row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
            for (name, disp, counter), v in per.items():
                acc[(name, counter)].append(v)
    for (name, counter), vals in sorted(acc.items()):
        pmc[name][counter] = sum(vals) / len(vals)
        totals[name][counter] = sum(vals)  # over every dispatch of the run: what a per-STEP figure is made from
        print(f"{d:10s} {name[:48]:48s} {counter:26s} n={len(vals)} mean={sum(vals) / len(vals):.6g}")

if dominant:
    import bench
    from raytracingrust_amd.build import kernel_sha

    args = bench.parse_args(bench_args)
    if args.config:
        from raytracingrust_amd.configs import baseline_config

        _, W, H = baseline_config(args.config, 1)
    else:
        W, H = args.width, args.height
    c = pmc.get(dominant, {})
    # every kernel of one bench step: HBM traffic is reported for the whole step (integrate + sum), VALU for the dominant one
    # (each kernel has its own number of dispatches per step -- the integrate kernel one per batch, k_sum_samples one per batch,
    # the streaming kernels one per depth: sum over every dispatch of the run, divided by the steps that were run)
    steps_run = max(args.steps + args.warmup, 1)
    fetch = sum(v.get("FETCH_SIZE", 0.0) for v in totals.values()) / steps_run
    write = sum(v.get("WRITE_SIZE", 0.0) for v in totals.values()) / steps_run
    calls_ratio = 1.0
    rec = {
        "_comment": "scripts/profile_bench.sh: rocprofv3 --kernel-trace --stats, then separate --pmc passes, of `python3 bench.py "
                    + " ".join(bench_args) + "`; means per dispatch of the dominant integrate kernel.  FETCH_SIZE / WRITE_SIZE are KiB "
                    "summed over the kernels of one step; FETCH_SIZE is doubled for hbm_bytes_per_launch as MI355X_MICROARCH.md "
                    "prescribes for gfx950 (128-B requests tallied at 64 B), WRITE_SIZE taken as is.",
        "kernel_sha": kernel_sha(),
        "workload": bench.workload_key(args, W, H),
        "command": "python3 bench.py " + " ".join(bench_args),
        "kernel": dominant,
        "kernel_avg_ns_rocprof": sum(durations[dominant]) / len(durations[dominant]),
        "kernel_min_ns_rocprof": min(durations[dominant]),
        "dispatches_per_step": len(durations[dominant]) / max(args.steps + args.warmup, 1),
        "vgpr": regs[dominant]["vgpr"], "sgpr": regs[dominant]["sgpr"], "scratch": regs[dominant]["scratch"],
        "sq_insts_valu": c.get("SQ_INSTS_VALU"), "sq_active_inst_valu": c.get("SQ_ACTIVE_INST_VALU"),
        "sq_thread_cycles_valu": c.get("SQ_THREAD_CYCLES_VALU"), "sq_wave_cycles": c.get("SQ_WAVE_CYCLES"),
        "sq_busy_cycles": c.get("SQ_BUSY_CYCLES"), "sq_waves": c.get("SQ_WAVES"), "sq_insts_salu": c.get("SQ_INSTS_SALU"),
        "sq_wait_any": c.get("SQ_WAIT_ANY"), "sq_wait_inst_any": c.get("SQ_WAIT_INST_ANY"), "sq_active_inst_any": c.get("SQ_ACTIVE_INST_ANY"),
        "sq_insts_lds": c.get("SQ_INSTS_LDS"), "sq_insts_vmem": c.get("SQ_INSTS_VMEM"),
        "fetch_size_kib": fetch, "write_size_kib": write,
        "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0 if (fetch or write) else None,
    }
    if rec["dispatches_per_step"] and rec["dispatches_per_step"] > 1.0:  # several dispatches of the dominant kernel per step (batches / depths)
        k = rec["dispatches_per_step"]
        for key in ("sq_insts_valu", "sq_active_inst_valu", "sq_thread_cycles_valu", "sq_wave_cycles", "sq_wait_any", "sq_wait_inst_any",
                    "sq_active_inst_any"):
            if rec[key] is not None:
                rec[key] *= k
        rec["kernel_avg_ns_rocprof"] *= k
        rec["_per_step_note"] = f"{k:g} dispatches of the dominant kernel per step: counters and time are per STEP (sum over them)"
    if rec["sq_thread_cycles_valu"] and rec["sq_active_inst_valu"]:
        rec["active_lane_fraction"] = rec["sq_thread_cycles_valu"] / (rec["sq_active_inst_valu"] * 64.0)
    if rec["sq_insts_valu"]:
        rec["valu_issue_frac_at_rocprof_time"] = rec["sq_insts_valu"] / (rec["kernel_avg_ns_rocprof"] * 1e-9) / (256 * 4 * 2.4e9 / 2)
    # what the compiler allocated for the dominant kernel (rocprofv3's VGPR_Count is in allocation granules)
    try:
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        import kernel_resources

        short = dominant.replace("void ", "").split("(")[0]
        for k in kernel_resources.resources():
            if k["name"].split("(")[0] == short:
                rec["compiled_vgprs"] = int(k.get("VGPRs", 0))
                rec["compiled_sgprs"] = int(k.get("TotalSGPRs", 0))
                rec["compiled_vgpr_spills"] = int(k.get("VGPRs Spill", 0))
                rec["compiled_sgpr_spills"] = int(k.get("SGPRs Spill", 0))
                rec["compiled_scratch_bytes_per_lane"] = int(k.get("ScratchSize [bytes/lane]", 0))
                rec["compiled_occupancy_waves_per_simd"] = int(k.get("Occupancy [waves/SIMD]", 0))
    except Exception as exc:  # the profile stands without it
        rec["compiled_resources_error"] = str(exc)
    path = os.path.join(out, rec["workload"] + ".json")
    with open(path, "w") as fh:
        json.dump(rec, fh, indent=1)
    print("== wrote", path)
    print(json.dumps(rec, indent=1))
