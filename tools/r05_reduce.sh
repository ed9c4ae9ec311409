#!/bin/bash
# split-K reduce kernel with four slabs in flight per thread: A/B against the previous library (_ab_head/libmoca_hip_base.so), same box, alternating
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/ab_reduce.txt
: > $OUT
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm_linear or gemm_conv or tconv or splitk" 2>&1 | tail -2 >> $OUT
bash tools/ab_lib2.sh _ab_head/libmoca_hip_base.so 2 "L3 " >> $OUT 2>&1
for i in 1 2 3; do
  for lib in _ab_head/libmoca_hip_base.so ""; do
    MOCA_HIP_LIB=$lib python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-video --no-emulate-world --no-fifo 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=${lib:-product}', 'UNet-steps/s', d['value'], 'ms_per_step', d['ms_per_step'])" >> $OUT
  done
done
cat $OUT
