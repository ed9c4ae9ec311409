#!/bin/bash
# virtual torch.cat (MOCA_VCAT): kernel tests, block / UNet parity, then same-box alternating A/B of the B = 2 step + FIFO iteration
mkdir -p gpurun_out/r05
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "two_source or virtual_cat or concat or colsum or gstat or upconv" > gpurun_out/r05/vcat_tests.txt 2>&1; echo "kernel tests rc=$?"; tail -3 gpurun_out/r05/vcat_tests.txt
timeout -k 10 900 python -m pytest tests/test_unet_gpu.py -x -q -m gpu > gpurun_out/r05/vcat_unet_tests.txt 2>&1; echo "unet tests rc=$?"; tail -3 gpurun_out/r05/vcat_unet_tests.txt
OUT=gpurun_out/r05/ab_vcat.txt
: > $OUT
for i in 1 2; do
  for v in 0 1; do
    MOCA_VCAT=$v python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-video --no-emulate-world 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('MOCA_VCAT=$v', 'UNet-steps/s', d['value'], 'ms_per_step', d['ms_per_step'], 'fifo_ms', d.get('fifo',{}).get('iteration_ms'), 'launches', d['roofline']['kernel'][:40])" >> $OUT
  done
done
cat $OUT
