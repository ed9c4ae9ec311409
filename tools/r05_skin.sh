#!/bin/bash
# split-K reduce inside the launch (moca_gemm_params.sk_counters; MOCA_SK_INKERNEL=0 switches the plan's use of it off): tests, per-launch and whole-step A/B
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/ab_skin.txt
: > $OUT
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "splitk or gemm_linear or gemm_conv or tconv" 2>&1 | tail -2 >> $OUT
timeout -k 10 600 python -m pytest tests/test_unet_gpu.py -x -q -m gpu 2>&1 | tail -1 >> $OUT
for i in 1 2; do
  for v in 0 1; do
    echo "== MOCA_SK_INKERNEL=$v" >> $OUT
    MOCA_SK_INKERNEL=$v python3 tools/plan_profile.py 2 2>/dev/null | grep -E "splits=" | cut -c1-150 >> $OUT
  done
done
for i in 1 2 3; do
  for v in 0 1; do
    MOCA_SK_INKERNEL=$v python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-video --no-emulate-world --no-fifo 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('MOCA_SK_INKERNEL=$v', 'UNet-steps/s', d['value'], 'ms_per_step', d['ms_per_step'])" >> $OUT
  done
done
cat $OUT
