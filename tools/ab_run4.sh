#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_ab4.txt; : > $O
for lib in "" tools/diag/libmoca_v1.so; do
  echo "== lib=${lib:-default}" >> $O
  MOCA_HIP_LIB=$lib python tools/bench_gemm.py "conv3x3 L0" "conv3x3 L1" "tconv3   L0" "tconv3   L1" "linear   L0 320->960" "linear   L1 2560" >> $O 2>&1
done
