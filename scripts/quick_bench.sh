#!/bin/bash
# parity tests + headline bench on the GPU box (through gpurun): bash scripts/quick_bench.sh [bench args]
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest.log 2>&1; tail -4 gpurun_out/pytest.log
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/bench.log 2>&1
python - <<'PY'
import json
for line in open('gpurun_out/bench.log'):
    if line.startswith('{'):
        d = json.loads(line)
        print("Mrays/s %.0f  ms/frame %.3f  kernel_ms %.3f  vgprs %s  frac %.3f" % (d['value'], d['ms_per_step'], d['config']['kernel_ms'], d['config']['vgprs'], d['roofline']['frac']))
        break
else:
    print(open('gpurun_out/bench.log').read()[-2000:])
PY
