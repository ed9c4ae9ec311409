#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_ab6.txt; : > $O
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "colsum or sweep or gemm" > gpurun_out/r2_t6.log 2>&1; tail -2 gpurun_out/r2_t6.log >> $O
for w in 1 2; do
  echo "== MOCA_GEMM_WIDE=$w" >> $O
  MOCA_GEMM_WIDE=$w python tools/bench_gemm.py "conv3x3 L0" "conv3x3 L1" "tconv3   L0" "tconv3   L1" >> $O 2>&1
done
for w in 1 2 1 2; do echo "bench WIDE=$w: $(MOCA_GEMM_WIDE=$w python bench.py --no-cpu-baseline --no-video --no-fifo 2>/dev/null | cut -c60-110)" >> $O; done
