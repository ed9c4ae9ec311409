#!/bin/bash
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/sqp2.txt
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
for B in 2 16; do
  for sh in geglu0 geglu1; do
    echo "=== BG_B=$B $sh sqp stamps (4th tile of every block)" >> $OUT
    BG_TUNE=7:1 BG_B=$B MOCA_HIP_DIAG=1 MOCA_HIP_LIB=tools/diag/libmoca_hip_stamps.so timeout -k 10 300 python tools/stamps.py $sh >> $OUT 2>&1
  done
done
