#!/bin/bash
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/g4q.txt
: > $OUT
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "g4p or lnfold" >> $OUT 2>&1 || { echo TESTS FAILED >> $OUT; exit 1; }
for rep in 1 2 3; do
  for cfg in "5:0" "5:1,6:0" "5:1,6:1"; do
    echo "=== rep $rep B=2 BG_TUNE=$cfg" >> $OUT
    BG_TUNE=$cfg BG_B=2 timeout -k 10 200 python tools/bench_gemm.py "geglu" >> $OUT 2>&1
  done
done
for cfg in "5:2,6:0" "5:2,6:1" "5:0"; do
  echo "=== B=2 all linears BG_TUNE=$cfg" >> $OUT
  BG_TUNE=$cfg BG_B=2 timeout -k 10 200 python tools/bench_gemm.py "linear" >> $OUT 2>&1
done
