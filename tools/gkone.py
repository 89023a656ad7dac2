"""Run ONE general-conv shape N times (for rocprofv3 --pmc via tools/pmc_one.sh): python tools/gkone.py [fwd|wgrad] [shape] [iters]
shapes: big = 3x3 466 -> 168 at 6 x 90 x 160 (GALD), small = 3x3 104 -> 104 at 16 x 22 x 22 (PraNet)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnd_semantic_segmentation_amd import gk
kind = sys.argv[1] if len(sys.argv) > 1 else "fwd"
shape = sys.argv[2] if len(sys.argv) > 2 else "big"
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
B, H, W, Ci, Co = (6, 90, 160, 466, 168) if shape == "big" else (16, 22, 22, 104, 104)
geom = (3, 3, 1, 1, 1, 1, 1, 1)
x = torch.randn((B, H, W, Ci), device="cuda").to(torch.bfloat16)
dy = torch.randn((B, H, W, Co), device="cuda").to(torch.bfloat16)
w = torch.randn((Co, Ci, 3, 3), device="cuda") * 0.05
wp, wpt = gk.gconv_pack(w)
dw = torch.empty_like(w)
for _ in range(iters):
    if kind == "fwd":
        gk.gconv(x, wp, Co, geom, stats=True)
    else:
        gk.gconv_wgrad(dy, x, dw, geom)
torch.cuda.synchronize()
if os.environ.get("GK_TIME"):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        if kind == "fwd":
            gk.gconv(x, wp, Co, geom, stats=True)
        else:
            gk.gconv_wgrad(dy, x, dw, geom)
    e1.record()
    torch.cuda.synchronize()
    print("%s %s: %.1f us per launch" % (kind, shape, e0.elapsed_time(e1) / 20 * 1e3))
