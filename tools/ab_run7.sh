#!/bin/bash
# whole-step A/B of switches on one device: $1 = env var name, values follow
cd $GRAFT_REPO_ROOT
VAR=$1; shift
O=gpurun_out/r2_ab_$VAR.txt; : > $O
for rep in 1 2; do for v in "$@"; do
  echo "== $VAR=$v" >> $O
  env $VAR=$v python bench.py --no-cpu-baseline --no-fifo --no-video --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])" >> $O
done; done
cat $O
