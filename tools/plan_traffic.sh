#!/bin/bash
# usage (GPU box, repo root): [PT_N=8] bash tools/plan_traffic.sh gpurun_out/traffic     (PT_N latents -> B = 2 PT_N videos)
set -e
OUT=$1
export TMPDIR=/tmp
mkdir -p $OUT
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 tools/plan_traffic.py run $OUT/steps.json > $OUT/fetch.log 2>&1
echo "fetch pass done" >> $OUT/progress.txt
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 tools/plan_traffic.py run $OUT/steps.json > $OUT/write.log 2>&1
echo "write pass done" >> $OUT/progress.txt
python3 tools/plan_traffic.py join $OUT > $OUT/plan_traffic.txt
rm -rf $OUT/fetch $OUT/write
head -30 $OUT/plan_traffic.txt
