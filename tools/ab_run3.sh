#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_ab3.txt; : > $O
for buf in 1 2; do
MOCA_GEMM_BUF=$buf python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k gemm > gpurun_out/r2_t2_$buf.log 2>&1; echo "BUF=$buf: $(tail -1 gpurun_out/r2_t2_$buf.log)" >> $O
done
for buf in 0 1 2; do
  echo "== MOCA_GEMM_BUF=$buf" >> $O
  MOCA_GEMM_BUF=$buf python tools/bench_gemm.py conv3x3 tconv3 "linear   L0" "linear   L1" "linear+res" >> $O 2>&1
done
