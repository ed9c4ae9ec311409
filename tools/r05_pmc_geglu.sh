#!/bin/bash
# fabric traffic + SQ counters of the 320-channel GEGLU launch at B = 16 / B = 2: round-4 kernels vs sqp vs g4p
export TMPDIR=/tmp
OUT=gpurun_out/r05/pmc_geglu
mkdir -p $OUT
for B in 16 2; do
for cfg in "5:0,7:0" "7:1" "5:1,7:0"; do
  tag=B${B}_$(echo $cfg | tr ':,' '__')
  mkdir -p $OUT/$tag; i=0
  for SET in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
    i=$((i+1))
    BG_TUNE=$cfg BG_B=$B rocprofv3 --pmc $SET --output-format csv -d $OUT/$tag/p$i -- python3 tools/bench_gemm.py "L0 320->2560 geglu" > $OUT/$tag/p$i.log 2>&1
  done
  python3 tools/pmc_summary.py $OUT/$tag gemm > $OUT/$tag.txt 2>&1
done
done
tail -n +1 $OUT/*.txt > gpurun_out/r05/pmc_geglu_summary.txt
