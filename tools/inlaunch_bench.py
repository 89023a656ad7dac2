"""Does the in-launch second-level reduction pay?  Per-launch times (HIP events over 200 back-to-back launches) of
  conv + BatchNorm finalize: mi_gconv (stats) + mi_gbn_finalize   vs   mi_gconv_bn (the conv's last workgroup finalizes)
  weight gradient:           mi_gconv_wgrad with its reducer launch vs   the (tap, tile)'s last workgroup adding the slabs
on the small-map shapes of PraNet's deep stages (16 x 22 x 22 / 16 x 11 x 11 / 16 x 44 x 44).   python tools/inlaunch_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnd_semantic_segmentation_amd import gk  # noqa: E402


def timeit(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    dev = torch.device("cuda")
    shapes = [(16, 22, 22, 104, 104, 3), (16, 22, 22, 1024, 416, 1), (16, 22, 22, 416, 1024, 1), (16, 11, 11, 208, 208, 3), (16, 11, 11, 2048, 832, 1), (16, 11, 11, 256, 256, 5),
              (16, 22, 22, 32, 32, 3), (16, 11, 11, 32, 32, 3)]
    print("%-34s %10s %10s   %10s %10s" % ("B H W Cin Cout k", "conv+fin", "conv_bn", "wgrad+red", "wgrad(in)"))
    for B, H, W, ci, co, k in shapes:
        x = torch.randn((B, H, W, ci), device=dev).to(torch.bfloat16)
        w = torch.randn((co, ci, k, k), device=dev) / (ci * k * k) ** 0.5
        wp, _ = gk.gconv_pack(w)
        geom = (k, k, 1, 1, k // 2, k // 2, 1, 1)
        gamma, beta = torch.ones(co, device=dev), torch.zeros(co, device=dev)
        rm, rv = torch.zeros(co, device=dev), torch.ones(co, device=dev)

        def two():
            y, st = gk.gconv(x, wp, co, geom, stats=True)
            gk.gbn_finalize(st, co, B * H * W, gamma, beta, rm, rv, 0.1, 1e-5)

        def one():
            gk.gconv_bn(x, wp, co, geom, gamma, beta, rm, rv, 0.1, 1e-5)
        dy = torch.randn((B, H, W, co), device=dev).to(torch.bfloat16)
        dw = torch.empty_like(w)
        res = [timeit(two), timeit(one)]
        for flag in (False, True):
            gk.INLAUNCH = flag
            res.append(timeit(lambda: gk.gconv_wgrad(dy, x, dw, geom)))
        gk.INLAUNCH = True
        print("%-34s %8.1f us %8.1f us   %8.1f us %8.1f us" % ((B, H, W, ci, co, k), *res))


if __name__ == "__main__":
    main()
