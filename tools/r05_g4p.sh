#!/bin/bash
# g4p (persistent two-blocks-per-CU kernel): parity tests, then A/B against the round-4 kernels on the GEGLU shapes
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/g4p.txt
: > $OUT
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "g4p or geglu or lnfold or sq256" >> $OUT 2>&1 || { echo TESTS FAILED >> $OUT; exit 1; }
for rep in 1 2; do
for B in 2 16; do
  echo "=== rep $rep B=$B g4p off (knob 5:0)" >> $OUT
  BG_TUNE=5:0 BG_B=$B timeout -k 10 200 python tools/bench_gemm.py "geglu" >> $OUT 2>&1
  echo "=== rep $rep B=$B g4p on" >> $OUT
  BG_TUNE=5:1 BG_B=$B timeout -k 10 200 python tools/bench_gemm.py "geglu" >> $OUT 2>&1
done
done
echo "=== B=2 every linear on g4p (5:2) vs default" >> $OUT
BG_TUNE=5:2 BG_B=2 timeout -k 10 200 python tools/bench_gemm.py "linear" >> $OUT 2>&1
BG_TUNE=5:0 BG_B=2 timeout -k 10 200 python tools/bench_gemm.py "linear" >> $OUT 2>&1
