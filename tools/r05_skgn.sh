#!/bin/bash
# split-K reduce inside the consuming GroupNorm (MOCA_SKGN): kernel test, UNet / block parity, same-box alternating A/B of the B = 2 step
mkdir -p gpurun_out/r05
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "splitk_groupnorm or groupnorm" > gpurun_out/r05/skgn_tests.txt 2>&1; echo "kernel tests rc=$?"; tail -3 gpurun_out/r05/skgn_tests.txt
timeout -k 10 900 python -m pytest tests/test_unet_gpu.py -x -q -m gpu > gpurun_out/r05/skgn_unet_tests.txt 2>&1; echo "unet tests rc=$?"; tail -3 gpurun_out/r05/skgn_unet_tests.txt
OUT=gpurun_out/r05/ab_skgn.txt
: > $OUT
for i in 1 2 3; do
  for v in 0 1 2; do
    MOCA_SKGN=$v python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-video --no-emulate-world --no-fifo 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('MOCA_SKGN=$v', 'UNet-steps/s', d['value'], 'ms_per_step', d['ms_per_step'], d['roofline']['kernel'][:44])" >> $OUT
  done
done
cat $OUT
