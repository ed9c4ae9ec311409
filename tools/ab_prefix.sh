#!/bin/bash
# same-device A/B of the shared CFG prefix: alternating runs of the headline bench
for i in 1 2 3; do
  for f in "" "--no-shared-prefix"; do
    python bench.py --steps 30 --no-fifo --no-video --no-cpu-baseline $f 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('shared' if d['config']['cfg_shared_prefix'] else 'plain ', d['value'], 'UNet-steps/s', d['roofline']['avg_launch_ms'], 'ms per launch')"
  done
done
