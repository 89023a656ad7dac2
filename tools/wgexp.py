"""Weight-gradient kernels at the BASELINE shapes: per-tap kernel (MI_WGRAD_P3=0) vs fused 3x3 rows (default), one process each."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnd_semantic_segmentation_amd import kernels as K  # noqa: E402
from tools.kbench import timeit  # noqa: E402

B, H = 8, 97
if len(sys.argv) > 1 and sys.argv[1] == "toggles":
    # phase toggles of the fused-row 3x3 kernel (wgrad_q3_kernel; experiment builds: MI355SEG_LIB=tools/experiments/libmi355seg_exp.so).  The library reads
    # MI_P3_DBG once per process, so every setting runs in a child process.
    import subprocess
    if len(sys.argv) > 2:
        ci = co = 256
        x = torch.randn((B, H, H, ci), device="cuda").to(torch.bfloat16)
        dy = torch.randn((B, H, H, co), device="cuda").to(torch.bfloat16)
        dw = torch.empty((co, ci, 3, 3), device="cuda")
        f = lambda: K.conv_wgrad(dy, x, dw, 3, 1, 2, 2)
        f()
        t = min(timeit(f, 20) for _ in range(3))
        print("q3 256 d2  %-22s %7.1f us (kernel + reducer)" % (sys.argv[2], t * 1e6), flush=True)
        sys.exit(0)
    for dbg, name in ((0, "all"), (8, "no stores"), (1, "no DMA"), (9, "no DMA, no stores"), (10, "no reads, no stores"), (12, "no MFMA, no stores"), (11, "MFMA only"), (14, "DMA only"),
                      (13, "reads only"), (15, "barriers only")):
        subprocess.run([sys.executable, os.path.abspath(__file__), "toggles", name], env=dict(os.environ, MI_P3_DBG=str(dbg)), check=True)
    sys.exit(0)
for ci, co, k, d in ((256, 256, 3, 2), (512, 512, 3, 4), (256, 256, 3, 1), (128, 128, 3, 1), (256, 1024, 1, 1), (1024, 256, 1, 1)):
    x = torch.randn((B, H, H, ci), device="cuda").to(torch.bfloat16)
    dy = torch.randn((B, H, H, co), device="cuda").to(torch.bfloat16)
    dw = torch.empty((co, ci, k, k), device="cuda")
    sc = torch.rand(co, device="cuda") + 0.5
    pad = d if k == 3 else 0
    f = lambda: K.conv_wgrad(dy, x, dw, k, 1, pad, d, scale=sc)
    f()
    t = min(timeit(f, 20) for _ in range(3))
    # the same launch with cold operands, as in the training step: 1 GiB written between two calls pushes x / dy out of the
    # 256 MB Infinity Cache; only the weight-gradient launches are timed (events around each call)
    if os.environ.get("COLD"):
        junk = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
        tc = []
        for _ in range(12):
            junk.add_(1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); f(); e1.record(); torch.cuda.synchronize()
            tc.append(e0.elapsed_time(e1) * 1e-3)
        tcold = sorted(tc)[len(tc) // 2]
        print("   cold operands: %7.1f us (median of 12)" % (tcold * 1e6))
        del junk
    err = float("nan")
    if os.environ.get("CHECK"):
        ref = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2).float(), dw.shape, dy.permute(0, 3, 1, 2).float(), padding=pad, dilation=d) * sc.view(-1, 1, 1, 1)
        err = ((dw - ref).abs().max() / ref.abs().max()).item()
    print("wgrad %dx%d %4d->%-4d d%d  P3=%s  %7.1f us  %5.0f TF  rel err %.2e" % (k, k, ci, co, d, os.environ.get("MI_WGRAD_P3", "1"), t * 1e6,
                                                                              2.0 * B * H * H * ci * co * k * k / t / 1e12, err))
