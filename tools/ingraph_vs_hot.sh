#!/bin/bash
# usage (GPU box, repo root): bash tools/ingraph_vs_hot.sh gpurun_out/ivh
set -e
OUT=$1
export TMPDIR=/tmp
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/g -- python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-fifo --no-video --no-emulate-world > $OUT/g.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/h -- python3 tools/plan_profile.py 2 > $OUT/h.log 2>&1
G=$(find $OUT/g -name "*kernel_trace.csv" | head -1)
H=$(find $OUT/h -name "*kernel_trace.csv" | head -1)
head -2 "$G" > $OUT/trace_header.txt
python3 tools/ingraph_vs_hot.py "$G" "$H" > $OUT/ingraph_vs_hot.txt
rm -rf $OUT/g $OUT/h
head -50 $OUT/ingraph_vs_hot.txt
