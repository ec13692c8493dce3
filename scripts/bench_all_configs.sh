#!/bin/bash
# Full-settings bench line for every single-GPU BASELINE config, reference trees and opt-in SAH trees.
# Output: gpurun_out/all_configs.jsonl (one bench.py JSON line per run).
mkdir -p gpurun_out
out=gpurun_out/all_configs.jsonl
: > $out
for c in c2 c2b c1 c3 c4 c5; do
    python bench.py --config $c --steps 3 --warmup 1 >> $out 2>gpurun_out/all_configs.err || exit 1
    echo "done $c reference"
    if [ $c != c2 ]; then
        python bench.py --config $c --steps 3 --warmup 1 --mesh-tree sah --scene-tree sah >> $out 2>>gpurun_out/all_configs.err || exit 1
        echo "done $c sah"
    fi
done
python - <<'PY'
import json
for l in open("gpurun_out/all_configs.jsonl"):
    d = json.loads(l)
    c = d["config"]
    print(c.get("workload", "")[:40], c.get("mesh_tree"), c.get("scene_tree"), "Mrays/s %.0f ms %.1f xCPU %.0f" % (d["value"], d["ms_per_step"], d["value"] / d["cpu_baseline"]["value"]))
PY
