"""Per-shape microbenchmark of the GEMM-class kernels at the BASELINE shapes (B=8, 769x769 -> 97x97 / 193x193).
Usage (GPU box):  python tools/kbench.py [--iters 20] [--only fwd,dgrad,wgrad]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnd_semantic_segmentation_amd import kernels as K  # noqa: E402

SHAPES = [  # name, H, Cin, Cout, k, dil, count per step
    ("l1 1x1 64->64", 193, 64, 64, 1, 1, 1), ("l1 3x3 64", 193, 64, 64, 3, 1, 3), ("l1 1x1 64->256", 193, 64, 256, 1, 1, 4),
    ("l1 1x1 256->64", 193, 256, 64, 1, 1, 2),
    ("l2 1x1 512->128", 97, 512, 128, 1, 1, 3), ("l2 3x3 128", 97, 128, 128, 3, 1, 3), ("l2 1x1 128->512", 97, 128, 512, 1, 1, 4),
    ("l3 1x1 1024->256", 97, 1024, 256, 1, 1, 22), ("l3 3x3 256 d2", 97, 256, 256, 3, 2, 22), ("l3 1x1 256->1024", 97, 256, 1024, 1, 1, 23),
    ("l4 1x1 2048->512", 97, 2048, 512, 1, 1, 2), ("l4 3x3 512 d4", 97, 512, 512, 3, 4, 2), ("l4 1x1 512->2048", 97, 512, 2048, 1, 1, 3),
    ("l4 1x1 1024->2048", 97, 1024, 2048, 1, 1, 1),
]


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="fwd,dgrad,wgrad,aspp")
    ap.add_argument("--batch", type=int, default=8)
    args = ap.parse_args()
    only = args.only.split(",")
    B = args.batch
    dev = "cuda"
    tot = {"fwd": [0, 0], "dgrad": [0, 0], "wgrad": [0, 0]}
    print("%-22s %10s %10s %10s   (TFLOP/s; ms)" % ("shape", "fwd", "dgrad", "wgrad"))
    for name, H, ci, co, k, d, cnt in SHAPES:
        x = torch.randn((B, H, H, ci), device=dev).to(torch.bfloat16)
        dy = torch.randn((B, H, H, co), device=dev).to(torch.bfloat16)
        w = torch.randn((co, ci, k, k), device=dev) * 0.05
        sc, sh = torch.rand(co, device=dev) + 0.5, torch.randn(co, device=dev)
        wp, wpt = K.pack_weight_fwd(w), K.pack_weight_dgrad(w, sc)
        dw = torch.empty_like(w)
        pad = d if k == 3 else 0
        flops = 2.0 * B * H * H * ci * co * k * k
        row = []
        for kind, fn in (("fwd", lambda: K.conv_gemm(x, wp, (H, H), k, 1, pad, d, scale=sc, bias=sh, relu=True)),
                         ("dgrad", lambda: K.conv_gemm(dy, wpt, (H, H), k, 1, pad, d, K.GATHER_DGRAD, msk=x)),
                         ("wgrad", lambda: K.conv_wgrad(dy, x, dw, k, 1, pad, d, scale=sc))):
            if kind not in only:
                row.append("-")
                continue
            t = timeit(fn, args.iters)
            tot[kind][0] += t * cnt
            tot[kind][1] += flops * cnt
            row.append("%6.0f %5.2f" % (flops / t / 1e12, t * 1e3))
        print("%-22s %12s %12s %12s   x%d" % (name, *row, cnt))
    if "aspp" in only:
        H, C = 97, 2048
        x = torch.randn((B, H, H, C), device=dev).to(torch.bfloat16)
        w4 = torch.randn((4, 19, C, 3, 3), device=dev) * 0.01
        wall, wallT = K.aspp_pack_fwd(w4), K.aspp_pack_dgrad(w4)
        g = torch.randn((B, H, H, K.ASPP_KPAD), device=dev).to(torch.bfloat16)
        dw4 = torch.empty_like(w4)
        fl = 2.0 * B * H * H * 684 * C
        t1 = timeit(lambda: K.conv_gemm(x, wall, (H, H), zsplit=K.ASPP_ZGW), args.iters)
        t2 = timeit(lambda: K.conv_gemm(g, wallT, (H, H), msk=x), args.iters)
        t3 = timeit(lambda: K.conv_wgrad(g, x, dw4, out_map=1, ncls=19), args.iters)
        print("%-22s %6.0f %5.2f %6.0f %5.2f %6.0f %5.2f" % ("aspp Z / dX / dW", fl / t1 / 1e12, t1 * 1e3, fl / t2 / 1e12, t2 * 1e3, fl / t3 / 1e12, t3 * 1e3))
        for kk, t in (("fwd", t1), ("dgrad", t2), ("wgrad", t3)):
            tot[kk][0] += t
            tot[kk][1] += fl
    for kk, (t, f) in tot.items():
        if t:
            print("TOTAL %-6s %8.2f ms/step  %7.1f TFLOP/s" % (kk, t * 1e3, f / t / 1e12))


if __name__ == "__main__":
    main()
