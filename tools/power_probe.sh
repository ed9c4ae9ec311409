#!/bin/bash
# Is the chip at its power cap during the hot kernels?  Samples `rocm-smi --showpower --showclocks` while one launch shape loops.
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/power_probe.txt
: > $OUT
rocm-smi --showmaxpower 2>/dev/null | grep -i "Max Graphics" >> $OUT
rocm-smi --showclocks 2>/dev/null | grep -iE "sclk|mclk" | head -2 >> $OUT
probe() {   # $1 = label, rest = command
  local label="$1"; shift
  echo "=== $label" >> $OUT
  ( "$@" > /tmp/pp.log 2>&1 ) &
  local PID=$!
  sleep 7
  for i in 1 2 3 4 5; do
    if kill -0 $PID 2>/dev/null; then
      rocm-smi --showpower --showclocks 2>/dev/null | grep -iE "Current Socket|sclk" | sed 's/GPU\[0\]\t\t: //' | tr '\n' ' ' >> $OUT
      echo >> $OUT
    fi
    sleep 1
  done
  wait $PID
  grep -E "us  |UNet-steps" /tmp/pp.log | tail -2 | cut -c1-200 >> $OUT
}
export BG_B=16
BG_ITERS=6000 probe "conv3x3 320->320, B=16 (K=2880, MFMA-bound: 1.0-1.1 PFLOP/s)" python3 tools/bench_gemm.py "conv3x3 L0 320->320"
BG_ITERS=5000 probe "GEGLU 320->2560, B=16 (persistent sqp kernel)" python3 tools/bench_gemm.py "linear   L0 320->2560"
BG_ITERS=20000 probe "linear+res 320->320, B=16 (HBM-bound)" python3 tools/bench_gemm.py "linear+res L0 320->320"
BG_PROBE=tattn BG_ITERS=3000 probe "q|k|v + temporal attention, B=16 (4 shapes in turn)" python3 tools/bench_gemm.py
probe "default bench step loop (B=2 DDIM-step graph), 400 steps" python3 bench.py --steps 400 --warmup 3 --no-cpu-baseline --no-video --no-emulate-world --no-fifo
cat $OUT
