#!/bin/bash
# gn_apply_gstat prologue (affine parameters fetched before the statistics hand-off): tests, per-launch (plan_profile, true operands) and whole-step A/B
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/ab_gnp.txt
: > $OUT
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "groupnorm or gstat or virtual_cat or concat" 2>&1 | tail -2 >> $OUT
for i in 1 2; do
  for lib in _ab_head/libmoca_hip_base.so ""; do
    echo "== lib=${lib:-product}" >> $OUT
    MOCA_HIP_LIB=$lib python3 tools/plan_profile.py 2 2>/dev/null | grep -E "groupnorm F=|splitk reduce" | cut -c1-140 >> $OUT
  done
done
for i in 1 2 3; do
  for lib in _ab_head/libmoca_hip_base.so ""; do
    MOCA_HIP_LIB=$lib python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-video --no-emulate-world 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=${lib:-product}', 'UNet-steps/s', d['value'], 'ms_per_step', d['ms_per_step'], 'fifo_ms', d.get('fifo',{}).get('iteration_ms'))" >> $OUT
  done
done
cat $OUT
