"""list the concat-related steps of the recorded B = 2 (shared prefix) forward: which sources of the virtual concats needed a statistics pass"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
dm = bench.build_model(dev, seed=321)
unet = dm.model.diffusion_model
B = int(os.environ.get("PP_B", "2"))
n = B // 2
x = torch.randn(n, 4, 16, 40, 64, device=dev)
ctx = torch.randn(B, 77, 1024, device=dev)
ts = torch.full((n,), 500, device=dev, dtype=torch.long)
with torch.no_grad():
    unet.forward_segments(x, ts, [ctx[:n], ctx[n:]], fps=torch.tensor([10] * n, device=dev), shared_x=True)
torch.cuda.synchronize()
plan = next(iter(unet._plans.values()))
for i, s in enumerate(plan.steps):
    fn, kw = s.func.__name__, s.keywords
    if fn in ("gstat_accum", "groupnorm_gstat_cat", "concat_channels_gstat", "concat_channels"):
        print(i, fn, {k: v for k, v in kw.items() if not torch.is_tensor(v)}, "merge_b" if fn == "groupnorm_gstat_cat" and s.args[6] is not None else "")
    elif fn == "gemm" and (kw.get("a2") is not None or (kw.get("gstat") is not None and len(kw["gstat"]) > 2)):
        g = kw.get("gstat")
        print(i, "gemm", "M", kw["M"], "N", s.args[1].N, "K", s.args[1].K, "a2" if kw.get("a2") is not None else "", "gstat", None if g is None else g[1:])
print(len(plan.steps), "steps")
