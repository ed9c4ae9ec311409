#!/bin/bash
# same-box A/B of whole-program settings: MOCA_TUNE values over rounds; B=2 step + FIFO iteration
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/ab_bench_$1.txt
: > $OUT
shift
for i in 1 2; do
  for v in "$@"; do
    MOCA_TUNE=$v python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-video --no-emulate-world 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('MOCA_TUNE=$v', 'UNet-steps/s', d['value'], 'ms_per_step', d['ms_per_step'], 'fifo_ms', d.get('fifo',{}).get('iteration_ms'))" >> $OUT
  done
done
cat $OUT
