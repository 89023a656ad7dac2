"""A/B of the two conv main loops on the shapes that matter (one process, interleaved rounds): mi_conv_gemm (128-wide tiles,
2 workgroups / CU) vs mi_conv_gemm_pp (320|256 x 256 tile, ping-pong wave groups).  Checks bit-equality of the outputs."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnd_semantic_segmentation_amd import kernels as K, _lib  # noqa: E402

B, H = 8, 97
L = _lib.lib()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None


def timeit(fn, iters):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def run(label, ci, co, k, d, flags, zg=0, rounds=3, iters=20):
    x = torch.randn((B, H, H, ci), device="cuda").to(torch.bfloat16)
    w = torch.randn((co, ci, k, k), device="cuda") * 0.05
    wp = K.pack_weight_fwd(w)
    f32 = bool(flags & 16)
    outs = [torch.zeros((B, H, H, co), device="cuda", dtype=torch.float32 if f32 else torch.bfloat16) for _ in range(3)]
    bits_in = torch.randint(-32768, 32767, (B, H, H, co // 16), device="cuda", dtype=torch.int16) if flags & 128 else None
    bits_out = [torch.zeros((B, H, H, co // 16), device="cuda", dtype=torch.int16) for _ in range(3)] if flags & 64 else [None] * 3
    sc = torch.rand(co, device="cuda") + 0.5
    sh = torch.randn(co, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    pad = d if k == 3 else 0

    def base(i=0, fl=flags):
        return L.mi_conv_gemm(P(x), P(wp), P(outs[i]), B, H, H, ci, H, H, co, k, 1, pad, d, 0, P(sc), P(sh), None, P(bits_in), P(bits_out[i]), fl, zg,
                              ctypes.c_float(0.0), st)

    def pp(mtg, i, fl=flags):
        return L.mi_conv_gemm_pp(P(x), P(wp), P(outs[i]), B, H, H, ci, H, H, co, k, 1, pad, d, 0, P(sc), P(sh), None, P(bits_in), P(bits_out[i]), fl, zg,
                                 ctypes.c_float(0.0), mtg, st)

    assert base(0) == 0, L.mi_last_error()
    assert pp(10, 1) == 0, L.mi_last_error()
    assert pp(8, 2) == 0, L.mi_last_error()
    torch.cuda.synchronize()
    eq10, eq8 = torch.equal(outs[0], outs[1]), torch.equal(outs[0], outs[2])
    if bits_out[0] is not None:
        eq10 = eq10 and torch.equal(bits_out[0], bits_out[1])
        eq8 = eq8 and torch.equal(bits_out[0], bits_out[2])
    if not (eq10 and eq8):
        d10 = (outs[0].float() - outs[1].float()).abs().max().item()
        d8 = (outs[0].float() - outs[2].float()).abs().max().item()
        print("  !! outputs differ: max|d| mtg10 %.3e mtg8 %.3e (|out| max %.3e)" % (d10, d8, outs[0].float().abs().max().item()))
    flops = 2.0 * B * H * H * ci * co * k * k
    res = {"base": [], "pp10": [], "pp8": [], "base-loop": [], "pp10-loop": []}
    pw_ok = k == 3 and flags in (69, 128)
    if pw_ok:
        outs.append(torch.zeros_like(outs[0]))
        bits_out.append(torch.zeros_like(bits_out[0]) if bits_out[0] is not None else None)
        assert pp(3, 3) == 0, L.mi_last_error()
        torch.cuda.synchronize()
        eqw = torch.equal(outs[0], outs[3]) and (bits_out[0] is None or torch.equal(bits_out[0], bits_out[3]))
        dw_ = (outs[0].float() - outs[3].float()).abs().max().item()
        print("  shared-window kernel (K order ky, c, kx instead of ky, kx, c): bit-equal %s, max|d| %.3e of |out| max %.3e" % (eqw, dw_, outs[0].float().abs().max().item()))
        res["pw"] = []
    nost = 1 << 30
    for _ in range(rounds):
        res["base"].append(timeit(lambda: base(0), iters))
        res["pp10"].append(timeit(lambda: pp(10, 1), iters))
        res["pp8"].append(timeit(lambda: pp(8, 2), iters))
        res["base-loop"].append(timeit(lambda: base(0, nost), iters))
        res["pp10-loop"].append(timeit(lambda: pp(10, 1, nost), iters))
        if pw_ok:
            res["pw"].append(timeit(lambda: pp(3, 3), iters))
    print("%-22s equal=%s/%s  " % (label, eq10, eq8) + "  ".join("%s %6.1f us %5.0f TF" % (n, min(v) * 1e6, flops / min(v) / 1e12) for n, v in res.items()))


if __name__ == "__main__":
    run("3x3 256 d2 fwd f69", 256, 256, 3, 2, 69)
    run("3x3 256 d2 dgrad f128", 256, 256, 3, 2, 128)
    run("3x3 512 d4 fwd f69", 512, 512, 3, 4, 69)
    run("1x1 1024->256 f69", 1024, 256, 1, 1, 69)
    run("1x1 2048->512 f69", 2048, 512, 1, 1, 69)
    run("1x1 1024->2048 f1", 1024, 2048, 1, 1, 1)
    run("aspp fwd 2048->720", 2048, 720, 1, 1, 48, zg=20)
    run("aspp dgrad 704->2048", 704, 2048, 1, 1, 0)
