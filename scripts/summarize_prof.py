#!/usr/bin/env python3
"""Condense rocprofv3 csv output (kernel stats + PMC counters) of scripts/profile_bench.sh into text tables."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


print("== kernel stats ==")
for f in find("trace/**/*kernel_stats.csv"):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            print({k: row[k] for k in row if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})
print("== kernel trace (k_render dispatches) ==")
for f in find("trace/**/*kernel_trace.csv"):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if "k_render" in row.get("Kernel_Name", ""):
                d = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
                print(row["Kernel_Name"][:60], "dur_ns", d, "vgpr", row.get("VGPR_Count"), "accum_vgpr", row.get("Accum_VGPR_Count"),
                      "sgpr", row.get("SGPR_Count"), "lds", row.get("LDS_Block_Size"), "scratch", row.get("Scratch_Size"),
                      "grid", row.get("Grid_Size"), "wg", row.get("Workgroup_Size"))
print("== PMC (per k_render dispatch: mean over dispatches) ==")
for d in ("pmc_sq1", "pmc_sq2", "pmc_fetch", "pmc_write"):
    acc = defaultdict(list)
    for f in find(d + "/**/*counter_collection.csv"):
        with open(f) as fh:
            per = defaultdict(float)
            for row in csv.DictReader(fh):
                if "k_render" not in row.get("Kernel_Name", ""):
                    continue
                per[(row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
            for (disp, name), v in per.items():
                acc[name].append(v)
    for name, vals in sorted(acc.items()):
        print(f"{d:10s} {name:28s} n={len(vals)} mean={sum(vals)/len(vals):.6g}")
