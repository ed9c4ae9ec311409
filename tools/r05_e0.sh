#!/bin/bash
# E0: g4 one-time skew of co-resident blocks; E1: GELU formulation (old = round 4)
set -e
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/e0.txt
: > $OUT
for rep in 1 2; do
for sk in 0 4000 8000 12000 16000 24000; do
  echo "=== rep $rep MOCA_G4_SKEW=$sk B=2 new gelu" >> $OUT
  MOCA_G4_SKEW=$sk BG_B=2 timeout -k 10 120 python tools/bench_gemm.py "geglu" >> $OUT 2>&1
done
echo "=== rep $rep old gelu B=2" >> $OUT
MOCA_HIP_DIAG=1 MOCA_HIP_LIB=tools/diag/libmoca_hip_gelu_old.so BG_B=2 timeout -k 10 120 python tools/bench_gemm.py "geglu" >> $OUT 2>&1
done
echo "=== B=16 new gelu" >> $OUT
BG_B=16 timeout -k 10 200 python tools/bench_gemm.py "geglu" >> $OUT 2>&1
echo "=== B=16 old gelu" >> $OUT
MOCA_HIP_DIAG=1 MOCA_HIP_LIB=tools/diag/libmoca_hip_gelu_old.so BG_B=16 timeout -k 10 200 python tools/bench_gemm.py "geglu" >> $OUT 2>&1
echo "=== B=16 new gelu, g4 forced (knob 1:2) skew 0 / 12000" >> $OUT
BG_TUNE=1:2,2:0 BG_B=16 timeout -k 10 200 python tools/bench_gemm.py "geglu" >> $OUT 2>&1
MOCA_G4_SKEW=12000 BG_TUNE=1:2,2:0 BG_B=16 timeout -k 10 200 python tools/bench_gemm.py "geglu" >> $OUT 2>&1
python -m pytest tests/test_kernels_gpu.py -x -q -k "geglu or sq256 or rowsum" -m gpu >> $OUT 2>&1 || true
