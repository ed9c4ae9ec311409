#!/usr/bin/env python3
"""Per-category summary of a rocprofv3 --kernel-trace --stats CSV of bench.py (forwards = warmup + steps)."""
import csv, glob, re, sys
f = sys.argv[1]
nfwd = float(sys.argv[2]) if len(sys.argv) > 2 else 6
rows = list(csv.DictReader(open(f)))
grp, cnt = {}, {}
for r in rows:
    n = r['Name']
    if 'splitk_gn' in n: k = 'splitk reduce + GroupNorm (one launch)'
    elif 'splitk_reduce' in n: k = 'splitk_reduce'            # (before the torch filter below: its 'reduce_kernel' pattern matched this kernel until round 5)
    elif 'gstat_accum' in n: k = 'statistics pass (virtual concat source)'
    elif any(x in n for x in ('distribution_elementwise', 'rocclr', 'FillFunctor', 'direct_copy', 'float16_copy', 'CatArray', 'index_', 'arange', 'gather_kernel', 'reduce_kernel', 'AbsFunctor', 'CompareEq', 'MulFunctor', 'CUDAFunctorOnSelf')):
        k = 'torch(init/host glue)'
    elif 'gemm_w80s' in n and re.search(r'<\d, 3>|ELi3EEE', n): k = 'gemm_w80s 320x192 (q|k|v + temporal attention)'
    elif 'gemm_w80s' in n and re.search(r'<\d, 2>|ELi2EEE', n): k = 'gemm_w80s 256x256 (wide GEGLU)'
    elif 'gemm_w80s' in n and re.search(r'<\d, 1>|ELi1EEE', n): k = 'gemm_w80s 160x320'
    elif 'gemm_w80s' in n: k = 'gemm_w80s 320x160'
    elif 'gemm_ws' in n: k = 'gemm_ws (weight-stationary 320->320 linears with a residual)'
    elif 'gemm_sqp' in n: k = 'gemm_sqp (persistent 256x256, register epilogue: GEGLU)'
    elif 'gemm_g4p' in n or 'gemm_g4q' in n: k = 'gemm_g4p (persistent 256x128)'
    elif 'gemm_w80' in n: k = 'gemm_w80/w80b (320x160)'
    elif 'gemm_g4' in n: k = 'gemm_g4 (GEGLU K<=640)'
    elif 'gemm_glds' in n: k = 'gemm_glds (256xBN)'
    elif 'gemm_f16' in n: k = 'gemm_small'
    elif 'temporal_attention' in n: k = 'temporal_attn'
    elif 'attention_v4' in n: k = 'attention_v4 (long keys)'
    elif 'attention_short' in n: k = 'attention_short (77-token context)'
    elif 'attention_kernel' in n: k = 'attention (short keys / causal)'
    elif 'gn_slab' in n: k = 'gn_slab (single launch)'
    elif 'gn_partial' in n: k = 'gn_partial'
    elif 'gn_apply_gstat' in n: k = 'gn_apply (statistics from the producer)'
    elif 'gn_apply' in n: k = 'gn_apply'
    elif 'concat_gstat' in n: k = 'concat + statistics'
    elif 'gn_finalize_colsum' in n: k = 'gn_finalize (from GEMM column sums)'
    elif 'gn_final' in n: k = 'gn_finalize'
    elif 'layernorm' in n: k = 'layernorm'
    else: k = 'other moca kernels'
    grp[k] = grp.get(k, 0) + float(r['TotalDurationNs']); cnt[k] = cnt.get(k, 0) + int(r['Calls'])
tot = sum(v for k, v in grp.items() if not k.startswith('torch'))
print(f"moca kernel time per forward: {tot/nfwd/1e6:.2f} ms")
for k, v in sorted(grp.items(), key=lambda x: -x[1]):
    if k.startswith('torch'):          # model build / weight fill / input staging: not launches of the timed graph replays
        continue
    print(f"{k:56s} {v/nfwd/1e6:7.2f} ms/fwd {100*v/tot:5.1f}%  launches/fwd {cnt[k]/nfwd:.0f}  avg {v/max(cnt[k],1)/1e3:8.1f} us")
glue = sum(v for k, v in grp.items() if k.startswith('torch'))
print(f"(not in the table: {sum(c for k, c in cnt.items() if k.startswith('torch'))} torch launches of model build / operand staging in the whole trace, "
      f"{glue/1e6:.2f} ms in total -- outside the timed graph replays; tracing inflates kernel durations by ~1.5 %, so the sum above is not an idle-time measurement)")
