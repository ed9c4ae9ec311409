#!/bin/bash
# A/B of the 256 x 256 staggered tiling (MOCA_GEMM_SQ256 = 0 / 1 / 2) on the wide projections, same device
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_ab8.txt; : > $O
for v in 0 1 2; do
  echo "== MOCA_GEMM_SQ256=$v" >> $O
  MOCA_GEMM_SQ256=$v python tools/bench_gemm.py "geglu" "->3840" "->1920" "linear   L0 320->960" >> $O 2>&1
done
