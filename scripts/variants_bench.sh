#!/bin/bash
# time kernel variants built with `python -m raytracingrust_amd.build --variant NAME FLAGS...` (timing only)
for v in "$@"; do
  RTR_HIP_LIBRARY=$PWD/raytracingrust_amd/_lib/librtr_hip_$v.so timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $BENCH_ARGS > gpurun_out/bench_$v.log 2>&1
  python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
for line in open(f'gpurun_out/bench_{v}.log'):
    if line.startswith('{'):
        d = json.loads(line)
        print("%-14s kernel_ms %.3f  segments %d  vgprs %s" % (v, d['config']['kernel_ms'], d['config']['segments_per_frame'], d['config']['vgprs']))
        break
else:
    print(v, open(f'gpurun_out/bench_{v}.log').read()[-1500:])
PY
done
