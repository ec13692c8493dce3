#!/bin/bash
# One rocprofv3 profile (kernel trace + PMC passes, scripts/profile_bench.sh) and one plain bench line per BASELINE
# configuration, at the kernel sources of this tree.  Through gpurun: bash scripts/profile_all.sh [round-tag]
# Afterwards, here: copy gpurun_out/profiles_<tag>/* into profiles/<tag>/ and commit.
set -u
TAG=${1:-r03}
CONFIGS=${CONFIGS:-"c2 c2b c1 c3 c4 c5"}
OUT=gpurun_out/profiles_$TAG
mkdir -p $OUT
touch $OUT/all_configs.jsonl
for c in $CONFIGS; do
    if [ $c = c2 ]; then args=""; else args="--config $c"; fi
    bash scripts/profile_bench.sh ${TAG}_$c $args > $OUT/profile_$c.log 2>&1
    cp gpurun_out/prof_${TAG}_$c/*.json $OUT/ 2>/dev/null
    cp gpurun_out/prof_${TAG}_$c/summary_tables.txt $OUT/${c}_summary.txt 2>/dev/null
    find gpurun_out/prof_${TAG}_$c -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${c}_kernel_stats.csv
    echo "profiled $c"
done
# bench lines AFTER the profiles are in place, so that each line carries its own counters
mkdir -p profiles/$TAG && cp $OUT/*.json profiles/$TAG/
for c in $CONFIGS; do
    if [ $c = c2 ]; then args=""; else args="--config $c"; fi
    python bench.py $args --steps 3 --warmup 1 >> $OUT/all_configs.jsonl 2>>$OUT/all_configs.err
    echo "benched $c"
done
python scripts/show_bench.py $OUT/all_configs.jsonl
