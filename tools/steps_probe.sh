#!/bin/bash
# does the measured step time depend on how many steps are timed / warmed up?  same box, alternating
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/steps_probe.txt
: > $OUT
for i in 1 2; do
  for sw in "20 3" "20 20" "100 10" "400 10"; do
    set -- $sw
    python3 bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-video --no-emulate-world --no-fifo 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps $1 warmup $2:', 'UNet-steps/s', d['value'], 'ms_per_step', d['ms_per_step'], 'avg_launch_ms', d['roofline']['avg_launch_ms'])" >> $OUT
  done
done
cat $OUT
