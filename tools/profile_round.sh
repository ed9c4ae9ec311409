#!/bin/bash
# One GPU-box pass that produces what profiles/ holds for a round (all on ONE box, named in box.txt):
#   kernel-trace stats of the bench command, the two PMC traffic passes, the per-step plan profile on TRUE operands (B = 2 and B = 16),
#   in-graph vs isolated per kernel, the in-bench SQ counters of the spatial attention kernel, the default bench line.
# usage (on the GPU box, repo root):  bash tools/profile_round.sh gpurun_out/prof_rNN
set -e
OUT=$1
export TMPDIR=/tmp
mkdir -p $OUT
{ hostname; rocm-smi --showuniqueid 2>/dev/null | grep -i "unique" | head -2; rocm-smi --showproductname 2>/dev/null | grep -i "series\|sku" | head -3; date -u; } > $OUT/box.txt 2>&1 || true
BENCH="bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fifo --no-video --no-emulate-world"   # 6 forwards of the B=2 (shared-prefix) plan
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktrace -- python3 $BENCH > $OUT/ktrace.log 2>&1
STATS=$(find $OUT/ktrace -name "*kernel_stats.csv" | head -1)
cp "$STATS" $OUT/kernel_stats.csv
python3 tools/prof_summary.py $OUT/kernel_stats.csv 6 > $OUT/kernel_stats_summary.txt
rm -rf $OUT/ktrace
echo "kernel stats done" >&2
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $BENCH > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $BENCH > $OUT/write.log 2>&1
python3 tools/pmc_traffic_summary.py $OUT 6 > $OUT/pmc_traffic_per_forward.txt
rm -rf $OUT/fetch $OUT/write
echo "traffic done" >&2
# SQ counters of attention_v4 INSIDE the bench (true operands; three passes, counters only)
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $OUT/pa$i -- python3 $BENCH > $OUT/pa$i.log 2>&1
done
python3 tools/pmc_summary.py $OUT attention_v4 > $OUT/pmc_attention_v4_in_bench.txt 2>&1
rm -rf $OUT/pa1 $OUT/pa2 $OUT/pa3
echo "attention pmc done" >&2
PP_JSON=$OUT/plan_profile_b2.json python3 tools/plan_profile.py 2 > $OUT/plan_profile_b2.txt 2>/dev/null
python3 tools/plan_profile.py 16 > $OUT/plan_profile_b16.txt 2>/dev/null
echo "plan profiles done" >&2
bash tools/ingraph_vs_hot.sh $OUT/ivh > /dev/null 2>&1 || true
cp $OUT/ivh/ingraph_vs_hot.txt $OUT/ingraph_vs_isolated.txt 2>/dev/null || true
python3 tools/bench_attn.py > $OUT/bench_attn.txt 2>&1
if [ -z "$SKIP_DEFAULT_BENCH" ]; then python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; fi
[ -f $OUT/bench_default.json ] && tail -c 400 $OUT/bench_default.json || true
