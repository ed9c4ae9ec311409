#!/bin/bash
# One GPU-box pass that produces everything profiles/ holds for a round:
#   kernel-trace stats of the bench command, the two PMC traffic passes, the per-step plan profile, the default bench line.
# usage (on the GPU box, repo root):  bash tools/profile_round.sh gpurun_out/prof_rNN
set -e
OUT=$1
export TMPDIR=/tmp
mkdir -p $OUT
BENCH="bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fifo --no-video"   # 6 forwards of the B=2 (shared-prefix) plan
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktrace -- python3 $BENCH > $OUT/ktrace.log 2>&1
STATS=$(find $OUT/ktrace -name "*kernel_stats.csv" | head -1)
cp "$STATS" $OUT/kernel_stats.csv
python3 tools/prof_summary.py $OUT/kernel_stats.csv 6 > $OUT/kernel_stats_summary.txt
rm -rf $OUT/ktrace
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $BENCH > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $BENCH > $OUT/write.log 2>&1
python3 tools/pmc_traffic_summary.py $OUT 6 > $OUT/pmc_traffic_per_forward.txt
rm -rf $OUT/fetch $OUT/write
python3 tools/plan_profile.py 2 > $OUT/plan_profile_b2.txt 2>/dev/null
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -1 $OUT/bench_default.json
