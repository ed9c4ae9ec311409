#!/bin/bash
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/sqp3.txt
: > $OUT
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "g4p" >> $OUT 2>&1 || { echo TESTS FAILED >> $OUT; exit 1; }
for rep in 1 2; do
  for B in 2 16; do
  for lib in product strict nont strict_nont; do
    echo "=== rep $rep B=$B sqp lib=$lib" >> $OUT
    if [ $lib = product ]; then BG_TUNE=7:1 BG_B=$B timeout -k 10 200 python tools/bench_gemm.py "geglu" >> $OUT 2>&1
    else MOCA_HIP_DIAG=1 MOCA_HIP_LIB=tools/diag/libmoca_hip_$lib.so BG_TUNE=7:1 BG_B=$B timeout -k 10 200 python tools/bench_gemm.py "geglu" >> $OUT 2>&1; fi
  done
  echo "=== rep $rep B=$B old kernels" >> $OUT
  BG_TUNE=7:0,5:0 BG_B=$B timeout -k 10 200 python tools/bench_gemm.py "geglu" >> $OUT 2>&1
  done
done
