#!/bin/bash
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/stamps_g4p.txt
: > $OUT
for B in 2 16; do
  for sh in geglu0 geglu1; do
    echo "=== BG_B=$B $sh g4p (stamps build; third tile of every block: phases = [0,0,main loop, stat publish+barrier, epilogue])" >> $OUT
    BG_TUNE=5:1 BG_B=$B MOCA_HIP_DIAG=1 MOCA_HIP_LIB=tools/diag/libmoca_hip_stamps.so timeout -k 10 300 python tools/stamps.py $sh >> $OUT 2>&1
  done
done
