#!/bin/bash
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/seg3.txt
: > $OUT
for B in 2; do
 for w in w0 w4; do
    echo "=== BG_B=$B geglu0 sqp, segments of iteration 2 of tile 4, wave ${w#w}" >> $OUT
    BG_TUNE=7:1 BG_B=$B MOCA_HIP_DIAG=1 MOCA_HIP_LIB=tools/diag/libmoca_hip_stamps_$w.so timeout -k 10 300 python tools/stamps.py geglu0 >> $OUT 2>&1
 done
done
