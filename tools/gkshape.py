"""General-conv launches at the PraNet / GALD shapes, one by one (each captured 40x in a HIP graph, operands rotated over 4 buffers so that a launch
does not find its own tile in the L2): us per launch, K steps, ns per K step.  usage: python tools/gkshape.py [pranet|gald] [fwd|dgrad|wgrad,...]
Kernel switches (MI_GCONV_KC, MI_GCONV_DEEP_WGS, MI_GCONV_REMAP, ...) come from the environment."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnd_semantic_segmentation_amd import gk  # noqa: E402

PRANET = [  # name, B, H, W, Cin, Cout, (kh, kw), dil, count per step
    ("l1 1x1 64->104", 16, 88, 88, 64, 104, (1, 1), 1, 1), ("l1 3x3 26", 16, 88, 88, 26, 26, (3, 3), 1, 9), ("l1 1x1 104->256", 16, 88, 88, 104, 256, (1, 1), 1, 3),
    ("l1 1x1 256->104", 16, 88, 88, 256, 104, (1, 1), 1, 2),
    ("l2 1x1 512->208", 16, 44, 44, 512, 208, (1, 1), 1, 3), ("l2 3x3 52", 16, 44, 44, 52, 52, (3, 3), 1, 12), ("l2 1x1 208->512", 16, 44, 44, 208, 512, (1, 1), 1, 4),
    ("l3 1x1 1024->416", 16, 22, 22, 1024, 416, (1, 1), 1, 5), ("l3 3x3 104", 16, 22, 22, 104, 104, (3, 3), 1, 18), ("l3 1x1 416->1024", 16, 22, 22, 416, 1024, (1, 1), 1, 6),
    ("l4 1x1 2048->832", 16, 11, 11, 2048, 832, (1, 1), 1, 2), ("l4 3x3 208", 16, 11, 11, 208, 208, (3, 3), 1, 9), ("l4 1x1 832->2048", 16, 11, 11, 832, 2048, (1, 1), 1, 3),
    ("rfb2 1x1 512->32", 16, 44, 44, 512, 32, (1, 1), 1, 5), ("rfb2 1x3 32", 16, 44, 44, 32, 32, (1, 3), 1, 3), ("rfb2 3x3 d3 32", 16, 44, 44, 32, 32, (3, 3), 3, 3),
    ("rfb2 cat 3x3 128->32", 16, 44, 44, 128, 32, (3, 3), 1, 1), ("rfb4 1x1 2048->32", 16, 11, 11, 2048, 32, (1, 1), 1, 5),
    ("ra 5x5 256->256", 16, 11, 11, 256, 256, (5, 5), 1, 3), ("ra 3x3 64->64", 16, 22, 22, 64, 64, (3, 3), 1, 4),
]
ALIGN = [  # the same conv with 16-byte aligned channel counts and with HarDNet's 4-byte aligned ones: what the unaligned 16-byte operand loads cost
    ("3x3 144->72 aligned", 6, 180, 320, 144, 72, (3, 3), 1, 1), ("3x3 142->68", 6, 180, 320, 142, 68, (3, 3), 1, 1),
    ("3x3 464->168 aligned", 6, 90, 160, 464, 168, (3, 3), 1, 1), ("3x3 466->168", 6, 90, 160, 466, 168, (3, 3), 1, 1),
]
GALD = [  # HarDNet-68 trunk at 6 x 720 x 1280 (bench.py --workload gald with MI_BENCH_SHAPES=1 lists them)
    ("hd 3x3 142->68 /4", 6, 180, 320, 142, 68, (3, 3), 1, 1), ("hd 3x3 102->40 /4", 6, 180, 320, 102, 40, (3, 3), 1, 1), ("hd 3x3 64->32 /2", 6, 360, 640, 64, 32, (3, 3), 1, 1),
    ("hd 3x3 466->168 /8", 6, 90, 160, 466, 168, (3, 3), 1, 1), ("hd 3x3 134->296 /8", 6, 90, 160, 296, 134, (3, 3), 1, 1), ("hd 3x3 368->98 /8", 6, 90, 160, 368, 98, (3, 3), 1, 1),
    ("hd 3x3 218->78 /8", 6, 90, 160, 218, 78, (3, 3), 1, 1), ("hd 3x3 740->334 /16", 6, 45, 80, 740, 334, (3, 3), 1, 1), ("hd 3x3 462->1072 /32", 6, 22, 40, 462, 1072, (3, 3), 1, 1),
    ("hd 1x1 262->256 /8", 6, 90, 160, 262, 256, (1, 1), 1, 1), ("hd 3x3 124->48 /4", 6, 180, 320, 124, 48, (3, 3), 1, 1), ("hd 3x3 24->14 /4", 6, 180, 320, 24, 14, (3, 3), 1, 1),
]


def timed(fn, n=40):
    for _ in range(3):
        fn(0)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(0)
        with torch.cuda.graph(g, stream=s):
            for i in range(n):
                fn(i)
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "pranet"
    kinds = (sys.argv[2] if len(sys.argv) > 2 else "fwd,dgrad,wgrad").split(",")
    tot = {k: 0.0 for k in kinds}
    print("%-28s %6s %5s | %s" % ("shape", "M", "steps", " | ".join("%-16s" % (k + " us (ns/step)") for k in kinds)))
    for name, B, H, W, ci, co, (kh, kw), d, cnt in {"pranet": PRANET, "gald": GALD, "align": ALIGN}[which]:
        geom = (kh, kw, 1, 1, d * (kh // 2), d * (kw // 2), d, d)
        R = 4
        xs = [torch.randn((B, H, W, ci), device="cuda").to(torch.bfloat16) for _ in range(R)]
        dys = [torch.randn((B, H, W, co), device="cuda").to(torch.bfloat16) for _ in range(R)]
        w = torch.randn((co, ci, kh, kw), device="cuda") * 0.05
        wp, wpt = gk.gconv_pack(w)
        dw = torch.empty_like(w)
        outs = [torch.empty((B, H, W, co), device="cuda", dtype=torch.bfloat16) for _ in range(R)]
        dxs = [torch.empty((B, H, W, ci), device="cuda", dtype=torch.bfloat16) for _ in range(R)]
        steps = kh * kw * ((ci + 63) // 64 if ci >= 64 else (ci + 31) // 32)
        cells = []
        for k in kinds:
            if k == "fwd":
                t = timed(lambda i: gk.gconv(xs[i % R], wp, co, geom, out=outs[i % R], stats=True))
            elif k == "dgrad":
                t = timed(lambda i: gk.gconv(dys[i % R], wpt, ci, geom, out=dxs[i % R], mode=gk.GATHER_DGRAD, out_hw=(H, W)))
            else:
                t = timed(lambda i: gk.gconv_wgrad(dys[i % R], xs[i % R], dw, geom))
            tot[k] += t * cnt
            cells.append("%7.1f (%5.0f)   " % (t, t * 1e3 / steps))
        print("%-28s %6d %5d | %s x%d" % (name, B * H * W, steps, " | ".join(cells), cnt))
    print("weighted totals (ms): " + ", ".join("%s %.2f" % (k, v * 1e-3) for k, v in tot.items()))


main()
