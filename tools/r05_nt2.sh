#!/bin/bash
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/nt2.txt
: > $OUT
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" >> $OUT 2>&1 || echo TESTS FAILED >> $OUT
for B in 2 16; do
    echo "=== B=$B product (size-gated nt residual loads)" >> $OUT
    BG_B=$B timeout -k 10 200 python tools/bench_gemm.py "linear+res" >> $OUT 2>&1
done
grep -v amdgpu.ids $OUT
