"""Per-shape cost of the BatchNorm statistics in the conv epilogue (mi_conv_gemm_stats) against the plain conv and the separate pass
(mi_bn_colsum2), at the BASELINE shapes (B = 8, 97 x 97 for layer3 / layer4, 193 x 193 for layer1).  usage: python tools/bn_shapes.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnd_semantic_segmentation_amd import kernels as K

def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

shapes = [(193, 64, 64, 1, 1), (193, 64, 64, 3, 1), (193, 64, 256, 1, 1), (193, 256, 64, 1, 1), (97, 512, 128, 1, 1), (97, 128, 128, 3, 1), (97, 128, 512, 1, 1),
          (97, 1024, 256, 1, 1), (97, 256, 256, 3, 2), (97, 256, 1024, 1, 1), (97, 2048, 512, 1, 1), (97, 512, 512, 3, 4), (97, 512, 2048, 1, 1)]
print("%-28s %9s %9s %9s %9s" % ("shape", "conv us", "st+fin us", "delta", "colsum2"))
for hw, ci, co, k, d in shapes:
    x = torch.randn(8, hw, hw, ci, device="cuda").to(torch.bfloat16)
    wp = K.pack_weight_fwd(torch.randn(co, ci, k, k, device="cuda") / (ci * k * k) ** 0.5)
    pilot = torch.zeros(co, device="cuda")
    pad = d * (k // 2)
    a = t(lambda: K.conv_gemm(x, wp, (hw, hw), k, 1, pad, d, K.GATHER_FWD))
    bn = torch.nn.BatchNorm2d(co).cuda()
    b = t(lambda: K.conv_gemm_stats(x, wp, (hw, hw), k, 1, pad, d, bn.running_mean, bn=bn))
    y = K.conv_gemm(x, wp, (hw, hw), k, 1, pad, d, K.GATHER_FWD)
    c = t(lambda: K.bn_colsum2(y, pilot))
    print("%-28s %9.1f %9.1f %9.1f %9.1f" % ("%dx%d %d->%d k%d d%d" % (hw, hw, ci, co, k, d), a, b, b - a, c))
