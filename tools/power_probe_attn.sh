#!/bin/bash
# power / clock while the 2560-token spatial self-attention loops
( python3 - <<'PY' > /tmp/pa.log 2>&1
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from moca_video_amd import ops
ops.set_stream(None)
F, heads, N = 32, 5, 2560
C = heads * 64
qkv = torch.randn(F * N, 3 * C, device="cuda").half()
out = torch.empty(F * N, C, device="cuda", dtype=torch.float16)
t0 = time.time()
n = 0
while time.time() - t0 < 14:
    for _ in range(200):
        ops.attention(qkv[:, :C], qkv[:, C:2*C], qkv[:, 2*C:], out, Bq=F, heads=heads, Nq=N, Nk=N, ldq=3*C, ldk=3*C, ldv=3*C, ldo=C, kv_div=1, scale=0.125)
    torch.cuda.synchronize(); n += 200
print("launches", n, "us each", (time.time() - t0) / n * 1e6)
PY
) &
PID=$!
sleep 8
for i in 1 2 3 4; do rocm-smi --showpower --showclocks 2>/dev/null | grep -iE "Current Socket|sclk" | sed 's/GPU\[0\]\t\t: //' | tr '\n' ' '; echo; sleep 1; done
wait $PID; cat /tmp/pa.log | tail -1
