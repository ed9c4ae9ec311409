#!/bin/bash
# same-device A/B of two builds of the library: tools/diag/libmoca_prev.so (the previous commit) against the tree's library.
# usage: bash tools/ab_lib.sh <script.py> [args]   (micro-benchmark)   |   bash tools/ab_lib.sh bench
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for lib in tools/diag/libmoca_prev.so ""; do
  echo "== lib=${lib:-tree}"
  if [ "$1" = "bench" ]; then
    MOCA_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-fifo --no-video --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
  else
    MOCA_HIP_LIB=$lib python "$@" 2>&1 | grep -v amdgpu.ids
  fi
done; done
