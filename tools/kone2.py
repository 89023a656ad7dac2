"""Run ONE 1x1 conv shape (layer3 conv3 forward: 256 -> 1024, FrozenBN + residual + ReLU + sign bits) N times for rocprofv3 --pmc."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnd_semantic_segmentation_amd import kernels as K
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
mode = sys.argv[2] if len(sys.argv) > 2 else "full"
B, H, ci, co = 8, 97, 256, 1024
x = torch.randn((B, H, H, ci), device="cuda").to(torch.bfloat16)
w = torch.randn((co, ci, 1, 1), device="cuda") * 0.05
wp = K.pack_weight_fwd(w)
r = torch.randn((B, H, H, co), device="cuda").to(torch.bfloat16)
bits = torch.empty((B, H, H, co // 16), device="cuda", dtype=torch.int16)
sc, sh = torch.rand(co, device="cuda") + 0.5, torch.randn(co, device="cuda")
for _ in range(iters):
    K.conv_gemm(x, wp, (H, H), scale=sc, bias=sh, res=r if mode == "full" else None, relu=True, mask_out=bits)
torch.cuda.synchronize()
