#!/bin/bash
# Same-box, alternating A/B of the two operating points every perf decision is taken on: the B = 2 DDIM-step graph
# (headline, UNet-steps/s) and the B = 16 FIFO iteration graph (ms).  Each variant is a string of environment assignments:
#   bash tools/ab_step.sh <out.txt> <rounds> "A: " "B: MOCA_TUNE=12:1" "C: MOCA_HIP_LIB=_ab_head/libmoca_hip_base.so"
# One line per run; copy the file you want judged to profiles/rNN_ab_<what>.txt.
OUT="$1"; N="$2"; shift 2
mkdir -p "$(dirname "$OUT")"; : > "$OUT"
for i in $(seq 1 "$N"); do
  for spec in "$@"; do
    name="${spec%%:*}"; envs="${spec#*:}"
    env $envs python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-video --no-emulate-world 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$name', '[$envs ]', 'UNet-steps/s', d['value'], 'ms_per_step', d['ms_per_step'], 'fifo_ms', d.get('fifo', {}).get('iteration_ms'))" >> "$OUT"
  done
done
cat "$OUT"
