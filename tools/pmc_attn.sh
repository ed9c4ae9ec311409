#!/bin/bash
# SQ counters of the spatial self-attention kernel at N=2560 (three --pmc passes, no tracing)
OUT=$1; shift
export TMPDIR=/tmp
mkdir -p $OUT
cat > /tmp/attn_one.py <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from moca_video_amd import ops
ops.set_stream(None)
F, heads, N = 32, 5, 2560
C = heads * 64
qkv = torch.randn(F * N, 3 * C, device="cuda").half()
out = torch.empty(F * N, C, device="cuda", dtype=torch.float16)
for _ in range(5):
    ops.attention(qkv[:, :C], qkv[:, C:2*C], qkv[:, 2*C:], out, Bq=F, heads=heads, Nq=N, Nk=N, ldq=3*C, ldk=3*C, ldv=3*C, ldo=C, kv_div=1, scale=0.125)
torch.cuda.synchronize()
PY
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $OUT/p$i -- python3 /tmp/attn_one.py > $OUT/p$i.log 2>&1
done
python3 tools/pmc_summary.py $OUT attention > $OUT/summary.txt 2>&1
