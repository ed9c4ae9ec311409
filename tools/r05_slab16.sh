#!/bin/bash
# fp16 split-K slabs on the 256-row kernel (MOCA_TUNE knob 9): parity of the full-width UNet against the reference goldens with fp32 / fp16 slabs, per-launch and whole-step A/B
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/ab_slab16.txt
: > $OUT
for v in 0 1; do
  echo "== MOCA_TUNE=9:$v  parity (tests/test_unet_gpu.py full width, tests/test_kernels_gpu.py split-K cases)" >> $OUT
  MOCA_TUNE=9:$v timeout -k 10 600 python -m pytest tests/test_unet_gpu.py -x -q -m gpu -s -k "full_width or reduced" 2>&1 | grep -E "parity\]|passed|failed" | cut -c1-160 >> $OUT
  MOCA_TUNE=9:$v timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm_linear or gemm_conv or tconv or splitk" 2>&1 | tail -1 >> $OUT
done
for i in 1 2; do
  for v in 0 1; do
    echo "== BG_TUNE=9:$v" >> $OUT
    BG_B=2 BG_TUNE=9:$v BG_ITERS=200 python3 tools/bench_gemm.py " L3 " 2>/dev/null | grep "s=[45]" >> $OUT
  done
done
for i in 1 2 3; do
  for v in 0 1; do
    MOCA_TUNE=9:$v python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-video --no-emulate-world --no-fifo 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('MOCA_TUNE=9:$v', 'UNet-steps/s', d['value'], 'ms_per_step', d['ms_per_step'])" >> $OUT
  done
done
cat $OUT
