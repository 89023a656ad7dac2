"""Does the BatchNorm backward (sums, then input gradient: both read g and y) run faster when its working set fits the 256 MB Infinity Cache?
[M, 1024] once (308 MB of reads per kernel) against two [M, 512] problems back to back (154 MB each), M = 8 x 97 x 97."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rnd_semantic_segmentation_amd import kernels as K
M = 8 * 97 * 97
def t(fn, n=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def make(C):
    g = torch.randn(1, 1, M, C, device="cuda").to(torch.bfloat16); y = torch.randn(1, 1, M, C, device="cuda").to(torch.bfloat16)
    mean = torch.zeros(C, device="cuda"); inv = torch.ones(C, device="cuda"); gam = torch.ones(C, device="cuda")
    return g, y, mean, inv, gam
big = make(1024)
halves = [make(512), make(512)]
junk = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
def bwd(p):
    g, y, mean, inv, gam = p
    db, dg = K.bn_bwd_colsums(g, y, mean, inv)
    return K.bn_bwd_apply(g, y, mean, inv, gam, db, dg, M)
def cold(fn):
    def run():
        junk.zero_()            # push everything out of the Infinity Cache
        fn()
    return run
base = t(cold(lambda: None))
print("flush alone %.1f us" % base)
print("[M,1024] sums + apply          : %.1f us" % (t(cold(lambda: bwd(big))) - base))
print("2 x [M,512] sums + apply       : %.1f us" % (t(cold(lambda: [bwd(h) for h in halves])) - base))
print("[M,1024] sums only %.1f, apply only %.1f" % (t(cold(lambda: K.bn_bwd_colsums(*big[:4]))) - base, t(cold(lambda: K.bn_bwd_apply(big[0], big[1], big[2], big[3], big[4], big[2], big[2], M))) - base))
q = make(256)
print("[M,256] sums + apply           : %.1f us" % (t(cold(lambda: bwd(q))) - base))
print("[M,256] sums only %.1f, apply only (cold) %.1f" % (t(cold(lambda: K.bn_bwd_colsums(*q[:4]))) - base, t(cold(lambda: K.bn_bwd_apply(q[0], q[1], q[2], q[3], q[4], q[2], q[2], M))) - base))
