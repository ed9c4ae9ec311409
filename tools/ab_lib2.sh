#!/bin/bash
# same-device A/B of the product library against a diagnostic build: bash tools/ab_lib2.sh <diag .so> rounds <bench_gemm filters...>
DIAG="$1"; N=${2:-2}; shift 2
for i in $(seq 1 $N); do
  echo "== product library"; python3 tools/bench_gemm.py "$@" 2>/dev/null
  echo "== $DIAG"; MOCA_HIP_LIB=$DIAG python3 tools/bench_gemm.py "$@" 2>/dev/null
done
