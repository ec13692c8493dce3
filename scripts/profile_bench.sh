#!/bin/bash
# Profiles one bench.py workload on the GPU box: kernel trace + stats, then PMC passes (each in its own run, never
# combined with trace domains).  Writes gpurun_out/prof_<tag>/{summary_tables.txt,<workload>.json}; the JSON carries the
# hash of the kernel sources it was taken on (raytracingrust_amd.build.kernel_sha) and is what bench.py reads from
# profiles/<round>/ -- copy both files there and commit them.
# Usage (through gpurun): bash scripts/profile_bench.sh <tag> [bench args]
set -u
TAG=${1:-r03}; shift || true
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-traversal-counters $*"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/trace.log" 2>&1
echo "trace rc=$?" >> "$OUT/summary.txt"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d "$OUT/pmc_sq1" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_sq1.log" 2>&1
echo "pmc_sq1 rc=$?" >> "$OUT/summary.txt"
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC --output-format csv -d "$OUT/pmc_sq2" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_sq2.log" 2>&1
echo "pmc_sq2 rc=$?" >> "$OUT/summary.txt"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_fetch.log" 2>&1
echo "pmc_fetch rc=$?" >> "$OUT/summary.txt"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_write.log" 2>&1
echo "pmc_write rc=$?" >> "$OUT/summary.txt"
cd "$ROOT"
python3 scripts/summarize_prof.py "$OUT" $ARGS > "$OUT/summary_tables.txt" 2>&1
cat "$OUT/summary.txt"; tail -45 "$OUT/summary_tables.txt"
# drop bulky raw files, keep csv
find "$OUT" -name "*.db" -delete
du -sh "$OUT"
