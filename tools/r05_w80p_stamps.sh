#!/bin/bash
# phase stamps: one-tile-per-block staggered kernel (knob 9 = 0) against its persistent form (9 = 1), B = 16 shapes
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/w80p_stamps2.txt
: > $OUT
for lib in stamps_i20; do
for sh in conv0 lin0; do
  for t in "9:0" "9:1"; do
    echo "=== BG_B=16 $sh BG_TUNE=$t $lib" >> $OUT
    BG_B=16 BG_TUNE=$t MOCA_HIP_DIAG=1 MOCA_HIP_LIB=tools/diag/libmoca_hip_$lib.so timeout -k 10 300 python tools/stamps.py $sh >> $OUT 2>&1 || exit 1
  done
done
done
cat $OUT
