#!/bin/bash
# same-device A/B of an environment variable over several values: bash tools/ab_env.sh VAR "v1 v2 v3" rounds [bench args]
VAR="$1"; VALS="$2"; N=${3:-2}; shift 3
ARGS="--steps 20 --warmup 3 --no-cpu-baseline --no-fifo --no-video --no-emulate-world $@"
for i in $(seq 1 $N); do
  for v in $VALS; do
    env $VAR=$v python3 bench.py $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', 'UNet-steps/s', d['value'], 'ms_per_step', d['ms_per_step'], 'graph_ms', d['roofline']['avg_launch_ms'])"
  done
done
