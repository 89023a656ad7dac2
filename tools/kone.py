"""Run ONE conv shape N times (for rocprofv3 --pmc): python tools/kone.py [fwd|wgrad] [iters]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnd_semantic_segmentation_amd import kernels as K
kind = sys.argv[1] if len(sys.argv) > 1 else "fwd"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
B, H, C, d = 8, 97, 256, 2
x = torch.randn((B, H, H, C), device="cuda").to(torch.bfloat16)
dy = torch.randn((B, H, H, C), device="cuda").to(torch.bfloat16)
w = torch.randn((C, C, 3, 3), device="cuda") * 0.05
sc, sh = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
wp = K.pack_weight_fwd(w)
dw = torch.empty_like(w)
for _ in range(iters):
    if kind == "fwd":
        K.conv_gemm(x, wp, (H, H), 3, 1, d, d, scale=sc, bias=sh, relu=True)
    else:
        K.conv_wgrad(dy, x, dw, 3, 1, d, d, scale=sc)
torch.cuda.synchronize()
