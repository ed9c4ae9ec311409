#!/bin/bash
# same-device A/B of a bench.py flag: alternating runs (A = without, B = with the flag), N rounds, one line per run
#   bash tools/ab_flag.sh "--no-weight-prefetch" 3 [extra bench args] > gpurun_out/ab.txt
FLAG="$1"; N=${2:-3}; shift 2
ARGS="--steps 20 --warmup 3 --no-cpu-baseline --no-fifo --no-video --no-emulate-world $@"
for i in $(seq 1 $N); do
  for v in A B; do
    if [ $v = A ]; then F=""; else F="$FLAG"; fi
    python3 bench.py $ARGS $F 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', repr('$F'), 'UNet-steps/s', d['value'], 'ms_per_step', d['ms_per_step'], 'graph_ms', d['roofline']['avg_launch_ms'])"
  done
done
