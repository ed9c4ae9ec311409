#!/bin/bash
# phase stamps of the short-K GEMM kernels (diagnostic build) + baseline timings on the product library
set -e
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/stamps.txt
: > $OUT
for B in 2 16; do
  for sh in geglu0 geglu1 linres0 qkv0 tconv0 ff2res0; do
    echo "=== BG_B=$B $sh (stamps build)" >> $OUT
    BG_B=$B MOCA_HIP_DIAG=1 MOCA_HIP_LIB=tools/diag/libmoca_hip_stamps.so timeout -k 10 300 python tools/stamps.py $sh >> $OUT 2>&1
  done
done
echo "=== baseline product library" >> $OUT
BG_B=2 timeout -k 10 300 python tools/bench_gemm.py linear >> $OUT 2>&1
BG_B=16 timeout -k 10 300 python tools/bench_gemm.py "linear   L0" "linear+res L0" "linear   L1 640->5120" >> $OUT 2>&1
