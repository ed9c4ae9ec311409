#!/bin/bash
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/walk.txt
: > $OUT
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "g4p or lnfold or sqp" >> $OUT 2>&1 || { echo TESTS FAILED >> $OUT; }
for rep in 1 2 3; do
  for B in 2 16; do
  for cfg in "7:1,8:0" "7:1,8:1"; do
    echo "=== rep $rep B=$B BG_TUNE=$cfg" >> $OUT
    BG_TUNE=$cfg BG_B=$B timeout -k 10 200 python tools/bench_gemm.py "geglu" >> $OUT 2>&1
  done
  done
done
grep -v amdgpu.ids $OUT | grep -v "L3 "
