"""Experiments on the memory-bound 1x1 shapes of layer3 (M = 75 272): where does the time go?"""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnd_semantic_segmentation_amd import kernels as K, _lib
from tools.kbench import timeit
B, H = 8, 97
L = _lib.lib()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None


def run(ci, co, label, res=True, k=1, d=1):
    x = torch.randn((B, H, H, ci), device='cuda').to(torch.bfloat16)
    w = torch.randn((co, ci, k, k), device='cuda') * 0.05
    wp = K.pack_weight_fwd(w)
    out = torch.empty((B, H, H, co), device='cuda', dtype=torch.bfloat16)
    r = torch.randn((B, H, H, co), device='cuda').to(torch.bfloat16)
    bits = torch.empty((B, H, H, co // 16), device='cuda', dtype=torch.int16)
    sc = torch.rand(co, device='cuda') + 0.5
    sh = torch.randn(co, device='cuda')
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    by_min = (x.numel() + out.numel() * (2 if res else 1)) * 2
    cases = [(1 | 4 | (2 if res else 0) | 64, 'full')]
    cases += [(1 | 4 | 64, 'no-res')] if res else []
    cases += [(1 << 30, 'nostore'), (128 | (2 if res else 0), 'dgrad-bits')]
    for fl, name in cases:
        zg = 0
        f = lambda: L.mi_conv_gemm(P(x), P(wp), P(out), B, H, H, ci, H, H, co, k, 1, d if k == 3 else 0, d, 0, P(sc), P(sh), P(r), P(bits) if fl & 128 else None, P(bits), fl, zg,
                                   ctypes.c_float(0.0), st)
        t = timeit(f, 30)
        print('%-16s %-8s MT=%s %7.1f us  %6.0f TF  %5.2f TB/s(min traffic)' % (label, name, os.environ.get("MI_IGEMM_MT", "auto") + "/" + os.environ.get("MI_IGEMM_BN", "auto"), t * 1e6,
              2.0 * B * H * H * ci * co * k * k / t / 1e12, by_min / t / 1e12))


run(256, 1024, '1x1 256->1024')
run(1024, 256, '1x1 1024->256', res=False)
run(256, 256, '3x3 256 d2', res=False, k=3, d=2)
run(512, 512, '3x3 512 d4', res=False, k=3, d=4)
run(512, 2048, '1x1 512->2048')
