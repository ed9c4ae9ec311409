#!/bin/bash
# ablation of the w80 main loop + g4-vs-w80 on short-K linears (one box, one call)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_ab1.txt; : > $O
for lib in "" tools/diag/libmoca_NO_DMA.so tools/diag/libmoca_NO_READ.so tools/diag/libmoca_BOTH.so; do
  echo "== lib=${lib:-default}" >> $O
  MOCA_HIP_LIB=$lib python tools/bench_gemm.py "conv3x3 L0 320->320" "conv3x3 L1 640->640" "tconv3   L0" "linear   L0 320->960" "linear   L2 5120" >> $O 2>&1
done
echo "== default dispatch, linear" >> $O
python tools/bench_gemm.py linear >> $O 2>&1
echo "== W80=0 G4=2 (g4 forced where N%128==0), linear" >> $O
MOCA_GEMM_W80=0 MOCA_GEMM_G4=2 python tools/bench_gemm.py linear >> $O 2>&1
echo "== W80=0 (glds), linear" >> $O
MOCA_GEMM_W80=0 MOCA_GEMM_G4=0 python tools/bench_gemm.py linear >> $O 2>&1
echo "== zero operands default" >> $O
BG_ZERO=1 python tools/bench_gemm.py "conv3x3 L0 320->320" "linear   L0" >> $O 2>&1
