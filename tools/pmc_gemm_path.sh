#!/bin/bash
# Counters of the L2 -> L1 (TA) -> LDS operand path for the GEMM kernels INSIDE the bench (true operands): separate --pmc passes
# (TA has two counter slots per pass on gfx950: more in one pass aborts rocprofv3).
# usage (GPU box, repo root): bash tools/pmc_gemm_path.sh gpurun_out/pmc_gemm
OUT=$1
export TMPDIR=/tmp
mkdir -p $OUT
BENCH="bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fifo --no-video --no-emulate-world"
i=0
for SET in "TA_BUSY_avr GRBM_GUI_ACTIVE" \
           "TA_BUFFER_READ_LDS_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $SET --output-format csv -d $OUT/p$i -- python3 $BENCH > $OUT/p$i.log 2>&1 || echo "pass $i ($SET) failed" >> $OUT/progress.txt
  echo "pass $i done" >> $OUT/progress.txt
done
python3 tools/pmc_summary.py $OUT gemm_w80s gemm_glds gemm_g4 > $OUT/pmc_gemm_path.txt 2>&1
rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 $OUT/p5
