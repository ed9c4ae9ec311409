import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import bench_ws as B
from moca_video_amd import ops
ops.set_stream(None)
for M in (655360, 81920):
    for res in (False, True):
        us, gbs = B.bench(M, res, False, True, 1)
        print(f"ablate={os.environ.get('MOCA_WS_ABLATE','0')} M={M} res={res} COLD: {us:.1f} us {gbs:.0f} GB/s", flush=True)
