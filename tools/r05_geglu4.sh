#!/bin/bash
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/geglu4.txt
: > $OUT
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "g4p or lnfold or sqp or geglu" >> $OUT 2>&1 || { echo TESTS FAILED >> $OUT; }
for rep in 1 2 3; do
  for B in 2 16; do
  for lib in product geglu2; do
    echo "=== rep $rep B=$B sqp lib=$lib" >> $OUT
    if [ $lib = product ]; then BG_TUNE=7:1 BG_B=$B timeout -k 10 200 python tools/bench_gemm.py "geglu" >> $OUT 2>&1
    else MOCA_HIP_DIAG=1 MOCA_HIP_LIB=tools/diag/libmoca_hip_$lib.so BG_TUNE=7:1 BG_B=$B timeout -k 10 200 python tools/bench_gemm.py "geglu" >> $OUT 2>&1; fi
  done
  done
done
grep -v amdgpu.ids $OUT | grep -v "L3 "
