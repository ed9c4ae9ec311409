#!/bin/bash
# micro-benchmark A/B of one env switch: $1 = env var, $2.. = values; remaining args after -- go to tools/bench_gemm.py
cd $GRAFT_REPO_ROOT
VAR=$1; shift
VALS=()
while [ "$1" != "--" ] && [ -n "$1" ]; do VALS+=("$1"); shift; done
shift
for rep in 1 2; do for v in "${VALS[@]}"; do
  echo "== $VAR=$v"
  env $VAR=$v python tools/bench_gemm.py "$@" 2>&1 | grep -v amdgpu.ids
done; done
