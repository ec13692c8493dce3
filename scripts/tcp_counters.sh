#!/bin/bash
# Vector-L1 behaviour of one bench.py workload: accesses, L2 requests, latency sums (own PMC passes, no trace domains).
# Usage (through gpurun): bash scripts/tcp_counters.sh <tag> [bench args]; prints per-kernel means.
set -u
TAG=${1:-tcp}; shift || true
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/tcp_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-traversal-counters $*"
cd /tmp
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum --output-format csv -d "$OUT/p1" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/p1.log" 2>&1
rocprofv3 --pmc TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum --output-format csv -d "$OUT/p2" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/p2.log" 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d "$OUT/p3" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/p3.log" 2>&1
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:40]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    if "k_render" not in k and "k_wf" not in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print("   %-34s n=%d mean=%.4g" % (c, len(v), sum(v) / len(v)))
PY
find "$OUT" -name "*.db" -delete
