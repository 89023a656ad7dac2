"""One general-conv forward shape N times, for tools/pmc_one.sh: python tools/gkone2.py B H W Cin Cout k [iters]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnd_semantic_segmentation_amd import gk
B, H, W, Ci, Co, k = [int(v) for v in sys.argv[1:7]]
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 6
geom = (k, k, 1, 1, k // 2, k // 2, 1, 1)
xs = [torch.randn((B, H, W, Ci), device="cuda").to(torch.bfloat16) for _ in range(3)]
w = torch.randn((Co, Ci, k, k), device="cuda") * 0.05
wp, wpt = gk.gconv_pack(w)
for i in range(iters):
    gk.gconv(xs[i % 3], wp, Co, geom, stats=True)
torch.cuda.synchronize()
