#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_ab2.txt; : > $O
for lib in "" tools/diag/libmoca_A_HOT.so tools/diag/libmoca_A_SKIP.so tools/diag/libmoca_NO_DMA.so; do
  echo "== lib=${lib:-default}" >> $O
  MOCA_HIP_LIB=$lib python tools/bench_gemm.py "conv3x3 L0 320->320" "conv3x3 L1 640->640" "conv3x3 L2 1280->1280" "tconv3   L0" "tconv3   L1" >> $O 2>&1
done
