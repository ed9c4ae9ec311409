#!/bin/bash
# Energy per launch shape: rocm-smi socket power while ONE launch shape loops for ~12 s (the only regime rocm-smi's ~1 Hz
# sampling resolves) x the launch time = joules per launch; B = 16 shapes of the FIFO iteration + the 2560-token attention launch.
# usage (GPU box, repo root): bash tools/energy_per_shape.sh gpurun_out/r06/energy.txt ; copy to profiles/r06_energy_per_shape.txt
OUT=${1:-gpurun_out/r06/energy.txt}
mkdir -p "$(dirname "$OUT")"; : > "$OUT"
probe() {   # $1 = label, rest = command printing "... <us> us ..." lines
  local label="$1"; shift
  ( "$@" > /tmp/ep.log 2>&1 ) &
  local PID=$!
  sleep 6
  local P=() C=()
  for i in 1 2 3 4; do
    if kill -0 $PID 2>/dev/null; then
      P+=("$(rocm-smi --showpower 2>/dev/null | grep -i "Current Socket" | grep -oE "[0-9]+\.[0-9]+" | tail -1)")
      C+=("$(rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | grep -oE "\([0-9]+Mhz\)" | tr -d '()Mhz' | head -1)")
    fi
    sleep 1
  done
  wait $PID
  local US=$(grep -oE "[0-9]+\.[0-9]+ us" /tmp/ep.log | tail -1 | grep -oE "[0-9]+\.[0-9]+")
  python3 - "$label" "$US" "${P[*]}" "${C[*]}" >> "$OUT" <<'PY'
import sys
label, us, P, C = sys.argv[1], float(sys.argv[2] or "nan"), [float(x) for x in sys.argv[3].split()], [float(x) for x in sys.argv[4].split()]
w = sum(P) / max(len(P), 1); mhz = sum(C) / max(len(C), 1)
print(f"{label:62s} {us:9.1f} us  {w:7.0f} W  {mhz:6.0f} MHz  {w * us * 1e-6:8.4f} J per launch")
PY
}
echo "# tools/energy_per_shape.sh: socket power (rocm-smi, mean of 4 readings) while one launch shape loops, x launch time; MI355X cap 1400 W" >> "$OUT"
export BG_B=16
BG_ITERS=8000 probe "conv3x3 320->320 M=655360 (K=2880)" python3 tools/bench_gemm.py "conv3x3 L0 320->320"
BG_ITERS=20000 probe "tconv 320 M=655360 (K=960)" python3 tools/bench_gemm.py "tconv3   L0"
BG_ITERS=6000 probe "GEGLU 320->2560 M=655360 (gemm_sqp)" python3 tools/bench_gemm.py "linear   L0 320->2560"
BG_ITERS=12000 probe "q|k|v 320->960 M=655360 (plain)" python3 tools/bench_gemm.py "linear   L0 320->960"
BG_ITERS=12000 probe "FF2 +res 1280->320 M=655360" python3 tools/bench_gemm.py "linear+res L0 1280->320"
BG_TUNE=10:0 BG_ITERS=25000 probe "lin +res 320->320 M=655360, tiled 160x320 kernel" python3 tools/bench_gemm.py "linear+res L0 320->320"
BG_TUNE=10:1 BG_ITERS=30000 probe "lin +res 320->320 M=655360, weight-stationary kernel" python3 tools/bench_gemm.py "linear+res L0 320->320"
BG_ITERS=6000 probe "conv3x3 1280->1280 M=40960 (K=11520)" python3 tools/bench_gemm.py "conv3x3 L2 1280->1280"
probe "attention_v4 2560 tokens, F=32, 5 heads" python3 tools/loop_attn.py
cat "$OUT"
