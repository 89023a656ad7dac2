import ctypes, os, subprocess, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "tilecopy.so")
L = ctypes.CDLL(so)
M, N = 75272, 1024
a = torch.randn(M, N, device="cuda").to(torch.bfloat16)
b = torch.empty_like(a)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
by = a.numel() * 2
for BM, BNc in ((160, 128), (128, 128), (128, 256)):
    for remap in (0, 10):
        r = []
        for mode, mult in ((0 + remap, 2), (1 + remap, 1), (2 + remap, 1)):
            us = t(lambda: L.run_tilecopy(P(a), P(b), M, N, BM, BNc, mode, 0, st))
            r.append("%s %6.1f us %5.2f TB/s" % (("copy", "read", "write")[mode % 10], us, mult * by / us / 1e6))
        print("tile %3dx%4d %s | " % (BM, BNc, "row-contiguous lanes" if remap else "epilogue lane pattern ") + " | ".join(r))
