#!/bin/bash
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/sqp.txt
: > $OUT
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "g4p or lnfold or geglu or sq256" >> $OUT 2>&1 || { echo TESTS FAILED >> $OUT; exit 1; }
for rep in 1 2; do
  for B in 2 16; do
  for cfg in "5:0,7:0" "5:1,7:0" "7:1"; do
    echo "=== rep $rep B=$B BG_TUNE=$cfg" >> $OUT
    BG_TUNE=$cfg BG_B=$B timeout -k 10 200 python tools/bench_gemm.py "geglu" >> $OUT 2>&1
  done
  done
done
for cfg in "7:2" "7:0,5:0"; do
  echo "=== B=2 all linears BG_TUNE=$cfg" >> $OUT
  BG_TUNE=$cfg BG_B=2 timeout -k 10 200 python tools/bench_gemm.py "linear" >> $OUT 2>&1
done
