#!/bin/bash
# same-device A/B of two TREES: _ab_head/ (an older commit unpacked with `git archive`, built in place; git-ignored) against the
# working tree.  bench.py of each tree, headline leg only.
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for t in _ab_head .; do
  echo "== tree=$t"
  (cd $t && python bench.py --no-cpu-baseline --no-fifo --no-video --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])")
done; done
