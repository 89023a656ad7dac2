"""ASPP data-gradient GEMM dX = G Wall (M = 75 272, K = 704, N = 2048) and the forward Z = X Wall^T (K = 2048, N = 720, fp32 tap planes): whole launch
against the main loop alone (flag bit 30 of the runtime-flag kernels skips the epilogue) - how much of these short-contraction launches is store time."""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rnd_semantic_segmentation_amd import kernels as K, _lib
from tools.kbench import timeit
B, H = 8, 97
L = _lib.lib()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for ci, co, label, flagsets in ((704, 2048, "dX  K=704 N=2048", ((0, "plain store"), (1 << 30, "main loop only"))),
                                (2048, 720, "Z   K=2048 N=720", ((48, "fp32 tap planes"), (0, "plain bf16 store"), (1 << 30, "main loop only"))),
                                (256, 256, "3x3 256 d2 (K=2304)", ((0, "plain store"), (1 << 30, "main loop only")))):
    k = 3 if "3x3" in label else 1
    x = torch.randn((B, H, H, ci), device="cuda").to(torch.bfloat16)
    wp = K.pack_weight_fwd(torch.randn((co, ci, k, k), device="cuda") * 0.05)
    out = torch.empty((B * H * H * co,), device="cuda", dtype=torch.float32)
    for fl, name in flagsets:
        f = lambda: L.mi_conv_gemm(P(x), P(wp), P(out), B, H, H, ci, H, H, co, k, 1, 2 if k == 3 else 0, 2 if k == 3 else 1, 0, None, None, None, None, None, fl, 20 if fl == 48 else 0,
                                   ctypes.c_float(0.0), st)
        t = timeit(f, 30)
        print("%-22s %-18s %7.1f us  %6.0f TFLOP/s" % (label, name, t * 1e6, 2.0 * B * H * H * ci * co * k * k / t / 1e12))
