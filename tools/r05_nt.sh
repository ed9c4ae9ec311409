#!/bin/bash
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/nt.txt
: > $OUT
for rep in 1 2 3; do
  for B in 2 16; do
  for lib in product nta ntres ntboth; do
    echo "=== rep $rep B=$B lib=$lib" >> $OUT
    if [ $lib = product ]; then BG_B=$B timeout -k 10 200 python tools/bench_gemm.py "linear+res L0" "linear+res L1 640->640" "linear   L0 320->320" >> $OUT 2>&1
    else MOCA_HIP_DIAG=1 MOCA_HIP_LIB=tools/diag/libmoca_hip_$lib.so BG_B=$B timeout -k 10 200 python tools/bench_gemm.py "linear+res L0" "linear+res L1 640->640" "linear   L0 320->320" >> $OUT 2>&1; fi
  done
  done
done
grep -v amdgpu.ids $OUT
