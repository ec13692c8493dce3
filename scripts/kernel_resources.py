#!/usr/bin/env python3
"""Registers, spills, scratch and LDS of every kernel in csrc/rtr_kernels.hip as hipcc compiles it for gfx950 with the
product's flags (-Rpass-analysis=kernel-resource-usage).  usage: python scripts/kernel_resources.py [-Dflag ...] [--json]"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracingrust_amd import build as B  # noqa: E402


def resources(defines=()):
    flags = [f for f in B.FLAGS if f not in ("-shared", "-fPIC")]
    cmd = [B._hipcc()] + flags + list(defines) + ["-c", os.path.join(B.CSRC, "rtr_kernels.hip"), "-o", "/dev/null",
                                                  "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    kernels, cur = [], None
    for line in out.splitlines():
        m = re.search(r"remark:\s+(.+?): (\S+) \[-Rpass-analysis", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2).strip()
        if k == "Function Name":
            name = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip() or v
            cur = {"name": re.sub(r"\(rtr::RenderParams\)|^void ", "", name)}
            kernels.append(cur)
        elif cur is not None:
            cur[k] = v
    return kernels


if __name__ == "__main__":
    ks = resources([a for a in sys.argv[1:] if a.startswith("-D")])
    if "--json" in sys.argv:
        print(json.dumps(ks, indent=1))
    else:
        for k in ks:
            print("%-58s vgpr %3s sgpr %3s spill v/s %s/%s scratch %s occupancy %s lds %s" % (
                k["name"][:58], k.get("VGPRs", "?"), k.get("TotalSGPRs", "?"), k.get("VGPRs Spill", "?"), k.get("SGPRs Spill", "?"),
                k.get("ScratchSize [bytes/lane]", "?"), k.get("Occupancy [waves/SIMD]", "?"), k.get("LDS Size [bytes/block]", "?")))
