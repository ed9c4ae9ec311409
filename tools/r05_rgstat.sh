#!/bin/bash
# split-K reduce that finishes the GroupNorm statistics (16-frame GroupNorms of the 5 x 8-latent level; MOCA_RGSTAT=0 switches it off): per-launch and whole-step A/B
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/ab_rgstat.txt
: > $OUT
for i in 1 2; do
  for v in 0 1; do
    echo "== MOCA_RGSTAT=$v" >> $OUT
    MOCA_RGSTAT=$v python3 tools/plan_profile.py 2 2>/dev/null | grep -E "HW=40 C=1280|M=  1280 N= 1280 K= 3840|M=  1280 N= 1280 K=11520 \+res" | cut -c1-150 >> $OUT
  done
done
for i in 1 2 3; do
  for v in 0 1; do
    MOCA_RGSTAT=$v python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-video --no-emulate-world --no-fifo 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('MOCA_RGSTAT=$v', 'UNet-steps/s', d['value'], 'ms_per_step', d['ms_per_step'])" >> $OUT
  done
done
cat $OUT
