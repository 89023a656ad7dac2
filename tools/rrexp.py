"""A/B of the two main loops of igemm_pp_kernel in one process, interleaved rounds: the two-group ping-pong loop (mtg 10 / 8) against the rolling
fragment ring (mtg 110 / 108).  Same tile, same LDS image, same order of the fp32 sums: the outputs must be EQUAL BIT FOR BIT.
  MI355SEG_LIB=tools/experiments/libmi355seg_exp.so python tools/rrexp.py [cold]   (the rolling loop exists in experiment builds only: tools/experiments/build.sh)
           cold: a 512 MB fill between launches (operands out of the Infinity Cache, as in the training step)"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnd_semantic_segmentation_amd import kernels as K, _lib  # noqa: E402

B, H = 8, 97
L = _lib.lib()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
COLD = len(sys.argv) > 1 and sys.argv[1] == "cold"
junk = torch.empty(512 << 20, dtype=torch.uint8, device="cuda") if COLD else None


def timeit(fn, iters):
    e0 = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    e1 = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    torch.cuda.synchronize()
    for i in range(iters):
        if COLD:
            junk.fill_(i & 255)
        e0[i].record()
        fn()
        e1[i].record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in zip(e0, e1))
    return t[len(t) // 2] * 1e-3


def run(label, ci, co, k, d, flags, zg=0, rounds=3, iters=20, variants=(10, 110, 8, 108)):
    x = torch.randn((B, H, H, ci), device="cuda").to(torch.bfloat16)
    w = torch.randn((co, ci, k, k), device="cuda") * 0.05
    wp = K.pack_weight_fwd(w)
    f32 = bool(flags & 16)
    outs = [torch.zeros((B, H, H, co), device="cuda", dtype=torch.float32 if f32 else torch.bfloat16) for _ in variants]
    bits_in = torch.randint(-32768, 32767, (B, H, H, co // 16), device="cuda", dtype=torch.int16) if flags & 128 else None
    bits_out = [torch.zeros((B, H, H, co // 16), device="cuda", dtype=torch.int16) if flags & 64 else None for _ in variants]
    sc = torch.rand(co, device="cuda") + 0.5
    sh = torch.randn(co, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    pad = d if k == 3 else 0

    def pp(mtg, i, fl=flags):
        rc = L.mi_conv_gemm_pp(P(x), P(wp), P(outs[i]), B, H, H, ci, H, H, co, k, 1, pad, d, 0, P(sc), P(sh), None, P(bits_in), P(bits_out[i]), fl, zg,
                               ctypes.c_float(0.0), mtg, st)
        assert rc == 0, L.mi_last_error()

    for i, v in enumerate(variants):
        pp(v, i)
    torch.cuda.synchronize()
    eq = [torch.equal(outs[0], o) and (bits_out[0] is None or torch.equal(bits_out[0], b)) for o, b in zip(outs, bits_out)]
    flops = 2.0 * B * H * H * ci * co * k * k
    res = {v: [] for v in variants}
    loop = {v: [] for v in variants}
    for _ in range(rounds):
        for i, v in enumerate(variants):
            res[v].append(timeit(lambda: pp(v, i), iters))
            if flags < 0 or True:
                loop[v].append(timeit(lambda: pp(v, i, 1 << 30), iters))
    print("%-24s equal=%s  " % (label, "/".join(str(e)[0] for e in eq)) +
          "  ".join("mtg%-3d %6.1f us %5.0f TF (loop %6.1f)" % (v, min(res[v]) * 1e6, flops / min(res[v]) / 1e12, min(loop[v]) * 1e6) for v in variants), flush=True)


if __name__ == "__main__":
    print("cold operands" if COLD else "back-to-back launches (operands warm in the Infinity Cache)")
    run("3x3 256 d2 fwd f69", 256, 256, 3, 2, 69)
    run("3x3 256 d2 dgrad f128", 256, 256, 3, 2, 128)
    run("3x3 512 d4 fwd f69", 512, 512, 3, 4, 69)
    run("1x1 1024->256 f69", 1024, 256, 1, 1, 69)
    run("1x1 2048->512 f69", 2048, 512, 1, 1, 69)
    run("1x1 1024->2048 f1", 1024, 2048, 1, 1, 1)
    run("aspp fwd 2048->720", 2048, 720, 1, 1, 48, zg=20)
    run("aspp dgrad 704->2048", 704, 2048, 1, 1, 0)
