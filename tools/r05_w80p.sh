#!/bin/bash
# persistent staggered kernel (MOCA_TUNE_GEMM_W80P, knob 9): parity tests, then same-box A/B on the UNet's shapes
O=gpurun_out/r05
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "w80p" > $O/w80p_tests.txt 2>&1
rc=$?; tail -3 $O/w80p_tests.txt
[ $rc -ne 0 ] && exit $rc
rm -f $O/w80p_ab.txt
SH='"conv3x3 L0" "conv3x3 L1" "tconv3   L0" "tconv3   L1" "linear   L0 320->320 " "linear   L0 320->960" "linear   L0 1280->320" "linear   L1 640->640 " "linear   L1 2560->640" "linear+res"'
for B in ${BS:-16}; do
  for t in "9:0" "9:1" "9:0" "9:1"; do
    echo "== B=$B BG_TUNE=$t" >> $O/w80p_ab.txt
    eval BG_B=$B BG_TUNE=$t timeout -k 10 300 python tools/bench_gemm.py $SH >> $O/w80p_ab.txt 2>&1 || exit 1
  done
done
python3 - <<'PY'
import re,collections
cols=collections.OrderedDict(); cur=None; n=collections.Counter()
for l in open('gpurun_out/r05/w80p_ab.txt'):
    if l.startswith('=='):
        cur=l.strip('= \n'); n[cur]+=1; cur=f'{cur} #{n[cur]}'; continue
    m=re.match(r'(.{46})\s+([\d.]+) us', l)
    if m: cols.setdefault(m.group(1).strip(), {})[cur]=float(m.group(2))
heads=[]
for v in cols.values():
    for k in v:
        if k not in heads: heads.append(k)
print(' '*46+' '.join(f'{h[-20:]:>20s}' for h in heads))
for k,v in cols.items(): print(f'{k:46s}'+' '.join(f'{v.get(h,0):20.1f}' for h in heads))
PY
