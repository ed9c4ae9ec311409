#!/bin/bash
# kernel-to-kernel gaps inside the replayed hipGraph: rocprofv3 --kernel-trace of a short bench run, then tools/gap_summary.py
set -e
OUT=$1
export TMPDIR=/tmp
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-fifo --no-video --no-emulate-world > $OUT/kt.log 2>&1
TR=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
python3 tools/gap_summary.py "$TR" > $OUT/gap_summary.txt
rm -rf $OUT/kt
cat $OUT/gap_summary.txt
