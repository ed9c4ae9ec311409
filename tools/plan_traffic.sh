#!/bin/bash
# usage (GPU box, repo root): bash tools/plan_traffic.sh gpurun_out/traffic
set -e
OUT=$1
export TMPDIR=/tmp
mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 tools/plan_traffic.py run $OUT/steps.json > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 tools/plan_traffic.py run $OUT/steps.json > $OUT/write.log 2>&1
python3 tools/plan_traffic.py join $OUT > $OUT/plan_traffic.txt
rm -rf $OUT/fetch $OUT/write
head -50 $OUT/plan_traffic.txt
