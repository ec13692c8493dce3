#!/usr/bin/env python3
"""Static instruction mix of an integrate kernel's loops, from the ISA hipcc emits for gfx950 with the product's flags
(no GPU needed).  usage: python scripts/isa_mix.py [mangled-kernel-name-prefix]   (default: the headline kernel
k_render<WORLD_LINEAR_SPHERES, 5, 0, false, PLAIN = true>).  Prints, for the kernel's largest loops, the instruction count by class: spill traffic
(v_readlane / v_writelane), compares and selects (the pairs gfx950 separates by wait states), transcendental and 32-bit
integer multiplies (quarter rate), scalar ALU by kind (exec-mask handling, branches, s_nop wait states), LDS and memory."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracingrust_amd import build as B  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "_ZN3rtr8k_renderILj0ELj5ELj0ELb0ELb1EEEvNS_12RenderParamsE"
flags = [f for f in B.FLAGS if f not in ("-shared", "-fPIC")]
with tempfile.TemporaryDirectory() as tmp:
    asm = os.path.join(tmp, "k.s")
    subprocess.run([B._hipcc()] + flags + ["--cuda-device-only", "-S", "-o", asm, os.path.join(B.CSRC, "rtr_kernels.hip")], check=True,
                   capture_output=True)
    lines = open(asm).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(name) and l.rstrip().endswith(":") or l.startswith(name + ":"))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
labels, ins = {}, []
for l in lines[start:end + 1]:
    t = l.strip()
    m = re.match(r"^(\.LBB[0-9_]+):", t)
    if m:
        labels[m.group(1)] = len(ins)
        continue
    if not t or t.startswith(";") or t.startswith("."):
        continue
    ins.append(t)


def klass(op):
    if op in ("v_readlane_b32", "v_writelane_b32"):
        return "spill traffic (v_readlane / v_writelane)"
    if op == "v_readfirstlane_b32":
        return "v_readfirstlane"
    if op.startswith("v_cmp"):
        return "vector compares"
    if op.startswith("v_cndmask"):
        return "vector selects"
    if op in ("v_sqrt_f32_e32", "v_sqrt_f32_e64", "v_rcp_f32_e32", "v_rcp_f32_e64", "v_rsq_f32_e32", "v_rcp_iflag_f32_e32"):
        return "transcendental (quarter rate)"
    if op in ("v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "v_mul_lo_i32", "v_mul_hi_i32"):
        return "32-bit integer multiplies (quarter rate)"
    if op.startswith("v_div_"):
        return "division helpers (v_div_scale / fmas / fixup)"
    if op.startswith("v_mov"):
        return "vector moves"
    if op.startswith("v_"):
        return "other vector ALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.split("_")[0] in ("global", "buffer", "flat", "scratch"):
        return "vector memory"
    if op == "s_nop":
        return "s_nop wait states"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_cbranch") or op == "s_branch":
        return "scalar branches"
    if "saveexec" in op or op in ("s_or_b64", "s_and_b64", "s_andn2_b64", "s_xor_b64", "s_orn2_b64", "s_nor_b64", "s_mov_b64", "s_cselect_b64"):
        return "scalar exec-mask / lane-mask logic"
    if op.startswith("s_load") or op.startswith("s_buffer"):
        return "scalar loads"
    if op.startswith("s_"):
        return "other scalar ALU"
    return "other"


spans = {}
for idx, t in enumerate(ins):
    op = t.split()[0]
    if op.startswith("s_cbranch") or op == "s_branch":
        tgt = t.split()[-1]
        if tgt in labels and labels[tgt] <= idx:
            a, b = spans.get(tgt, (labels[tgt], idx))
            spans[tgt] = (min(a, labels[tgt]), max(b, idx))
print(f"kernel {name}: {len(ins)} instructions, {len(spans)} loops")
for tgt, (a, b) in sorted(spans.items(), key=lambda x: -(x[1][1] - x[1][0]))[:3]:
    c = collections.Counter(klass(t.split()[0]) for t in ins[a:b + 1])
    total = sum(c.values())
    vec = sum(v for k, v in c.items() if k.startswith("vector") or k.startswith("other vector") or k.startswith("spill") or k.startswith("trans") or k.startswith("32-bit") or k.startswith("division") or k == "v_readfirstlane")
    sca = sum(v for k, v in c.items() if k.startswith("scalar") or k.startswith("other scalar") or k.startswith("s_"))
    print(f"\nloop {tgt}: {total} instructions (static), vector ALU {vec}, scalar {sca}")
    for k, v in c.most_common():
        print(f"    {k:52s} {v:5d}  {100.0 * v / total:5.1f} %")
