"""Times the chained 1x1 kernel (mi_conv_chain) against the two launches it replaces, at the training step's size (B = 8, 97 x 97: M = 75 272), forward
and backward, over rotating operand sets (so that nothing is served from the 256 MB Infinity Cache that the step would not find there).
usage: python tools/chainexp.py [iters] [grid ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnd_semantic_segmentation_amd import kernels as K  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_chain as T  # noqa: E402

T.K = K


def timeit(fn, sets, iters):
    for s in sets:
        fn(s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(sets[i % len(sets)])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    grids = [int(g) for g in sys.argv[2:]] or [0]
    B, H, W = 8, 97, 97
    M = B * H * W
    for backward in (False, True):
        sets = [T._operands(B, H, W, 11 + i, backward) for i in range(4)]
        outs = [(torch.empty(B, H, W, 1024, dtype=torch.bfloat16, device="cuda"), torch.empty(B, H, W, 256, dtype=torch.bfloat16, device="cuda"),
                 torch.empty(B, H, W, 64, dtype=torch.int16, device="cuda"), torch.empty(B, H, W, 16, dtype=torch.int16, device="cuda")) for _ in sets]
        idx = {id(s): o for s, o in zip(sets, outs)}

        def two(s):
            a, res, w1, w2, sc1, sh1, sc2, sh2, bits1, bits2 = s
            mid, out, b1, b2 = idx[id(s)]
            if backward:
                K.conv_gemm(a, w1, (H, W), res=res, bits=bits1, out=mid)
                K.conv_gemm(mid, w2, (H, W), bits=bits2, out=out)
            else:
                K.conv_gemm(a, w1, (H, W), scale=sc1, bias=sh1, res=res, relu=True, mask_out=b1, out=mid)
                K.conv_gemm(mid, w2, (H, W), scale=sc2, bias=sh2, relu=True, mask_out=b2, out=out)

        def first_only(s):
            a, res, w1, w2, sc1, sh1, sc2, sh2, bits1, bits2 = s
            mid, out, b1, b2 = idx[id(s)]
            if backward:
                K.conv_gemm(a, w1, (H, W), res=res, bits=bits1, out=mid)
            else:
                K.conv_gemm(a, w1, (H, W), scale=sc1, bias=sh1, res=res, relu=True, mask_out=b1, out=mid)

        t2 = timeit(two, sets, iters)
        t1 = timeit(first_only, sets, iters)
        alg = M * (256 + 1024 + 1024 + 256) * 2 + M * (128 + 32)
        print("%s: two launches %.1f us (first alone %.1f)" % ("backward" if backward else "forward", t2, t1), flush=True)
        for g in grids:
            def ch(s, g=g):
                a, res, w1, w2, sc1, sh1, sc2, sh2, bits1, bits2 = s
                mid, out, b1, b2 = idx[id(s)]
                if backward:
                    K.conv_chain(a, w1, res, w2, bits1=bits1, bits2=bits2, mid=mid, out=out, grid=g)
                else:
                    K.conv_chain(a, w1, res, w2, scale1=sc1, shift1=sh1, scale2=sc2, shift2=sh2, mid=mid, out=out, bits1_out=b1, bits2_out=b2, grid=g)
            tc = timeit(ch, sets, iters)
            print("   chain grid %3d: %.1f us  = %.2f TB/s of its %.0f MB, %.0f TFLOP/s" % (g, tc, alg / tc / 1e6, alg / 1e6, 2.0 * M * 1024 * 512 / tc / 1e6), flush=True)


if __name__ == "__main__":
    main()
