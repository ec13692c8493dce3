#!/bin/bash
# Counter comparison of option sets / library builds on one configuration (through gpurun):
#   bash scripts/pmc_compare.sh <tag> <config> <spp> "<opts>" [lib.so]
# Two rocprofv3 --pmc passes (never combined with trace domains) of scripts/option_bench.py, summarised per kernel.
set -u
TAG=$1; CONFIG=$2; SPP=$3; OPTS=$4; LIB=${5:-}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp REPEATS=2
if [ -n "$LIB" ]; then export RTR_HIP_LIBRARY=$ROOT/$LIB; fi
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d "$OUT/p1" -- python3 "$ROOT/scripts/option_bench.py" $CONFIG $SPP "$OPTS" > "$OUT/p1.log" 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC --output-format csv -d "$OUT/p2" -- python3 "$ROOT/scripts/option_bench.py" $CONFIG $SPP "$OPTS" > "$OUT/p2.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/tr" -- python3 "$ROOT/scripts/option_bench.py" $CONFIG $SPP "$OPTS" > "$OUT/tr.log" 2>&1
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for d in ("p1", "p2"):
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        per = defaultdict(float)
        for row in csv.DictReader(open(f)):
            per[(row["Kernel_Name"][:60], row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
        for (k, disp, c), v in per.items():
            acc[k][c].append(v)
for k in acc:
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-26s n=%d mean %.4e" % (c, len(v), sum(v) / len(v)))
for f in glob.glob(os.path.join(out, "tr", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        print("stats", row["Name"][:60], row["Calls"], row["AverageNs"])
PY
find "$OUT" -name "*.db" -delete
