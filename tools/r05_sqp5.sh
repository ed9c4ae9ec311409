#!/bin/bash
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/sqp5.txt
: > $OUT
for rep in 1 2 3; do
  for B in 2 16; do
  for lib in product v1 v2; do
    echo "=== rep $rep B=$B sqp lib=$lib" >> $OUT
    if [ $lib = product ]; then BG_TUNE=7:1 BG_B=$B timeout -k 10 200 python tools/bench_gemm.py "geglu" >> $OUT 2>&1
    else MOCA_HIP_DIAG=1 MOCA_HIP_LIB=tools/diag/libmoca_hip_$lib.so BG_TUNE=7:1 BG_B=$B timeout -k 10 200 python tools/bench_gemm.py "geglu" >> $OUT 2>&1; fi
  done
  done
done
