#!/bin/bash
# batched (one B=2 graph) vs concurrent (two B=1 graphs on two streams) CFG step, same device
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_ab9.txt; : > $O
for m in batched concurrent batched concurrent; do
  echo "== cfg-mode $m" >> $O
  python bench.py --no-cpu-baseline --no-fifo --no-video --cfg-mode $m --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])" >> $O
done
