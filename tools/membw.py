"""Achievable HBM bandwidth on this GPU for the access mixes the 1x1 convs have (copy, add, read-only)."""
import torch
M, N = 75272, 1024
a = torch.randn(M, N, device="cuda").to(torch.bfloat16)
b = torch.randn(M, N, device="cuda").to(torch.bfloat16)
c = torch.empty_like(a)
def t(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
by = a.numel() * 2
us = t(lambda: c.copy_(a)); print("copy   %.1f us  %.2f TB/s" % (us, 2 * by / us / 1e6))
us = t(lambda: torch.add(a, b, out=c)); print("add    %.1f us  %.2f TB/s" % (us, 3 * by / us / 1e6))
us = t(lambda: torch.relu_(c)); print("relu_  %.1f us  %.2f TB/s" % (us, 2 * by / us / 1e6))
us = t(lambda: a.sum()); print("sum    %.1f us  %.2f TB/s" % (us, by / us / 1e6))
big = torch.empty(1 << 30, dtype=torch.uint8, device="cuda"); big2 = torch.empty_like(big)
us = t(lambda: big2.copy_(big)); print("copy 1GiB %.1f us  %.2f TB/s" % (us, 2 * (1 << 30) / us / 1e6))
us = t(lambda: big.zero_()); print("memset 1GiB %.1f us  %.2f TB/s" % (us, (1 << 30) / us / 1e6))
