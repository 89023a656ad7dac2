"""Timeline of the ping-pong conv main loop (igemm_pp_kernel) from s_memtime stamps: builds a second library with -DMI_PP_TRACE
(`python tools/pptrace.py build`, in the build container), then on the GPU box prints, per wave group, the average cycles between
the six stamps of a k-step:  0 top | 1 DMA issued | 2 fragments read (lgkmcnt 0) | 3 past barrier 1 | 4 MFMAs issued | 5 past barrier 2."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tools", "experiments", "libpptrace.so")
sys.path.insert(0, ROOT)


def build(extra=()):
    """extra: more -D flags; the library then gets a suffix (PPTRACE_SO selects it at run time)."""
    import __graft_entry__ as g
    csrc = os.path.join(ROOT, "rnd_semantic_segmentation_amd", "csrc")
    so = SO if not extra else SO.replace(".so", "_" + "_".join(e.replace("=", "") for e in extra) + ".so")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-result", "-DMI_PP_TRACE",
           "-I", os.path.join(ROOT, "include"), "-o", so] + ["-D" + e for e in extra] + g.SOURCES + ["-ldl"]
    subprocess.run(cmd, check=True, cwd=csrc)
    print(so)


def main():
    import torch
    from rnd_semantic_segmentation_amd import kernels as K
    L = ctypes.CDLL(os.environ.get("PPTRACE_SO", SO))
    B, H = 8, 97
    ci = co = int(os.environ.get("C", "256"))
    d = 2
    x = torch.randn((B, H, H, ci), device="cuda").to(torch.bfloat16)
    w = torch.randn((co, ci, 3, 3), device="cuda") * 0.05
    wp = K.pack_weight_fwd(w)
    out = torch.zeros((B, H, H, co), device="cuda", dtype=torch.bfloat16)
    bits = torch.zeros((B, H, H, co // 16), device="cuda", dtype=torch.int16)
    sc, sh = torch.rand(co, device="cuda") + 0.5, torch.randn(co, device="cuda")
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        rc = L.mi_conv_gemm_pp(P(x), P(wp), P(out), B, H, H, ci, H, H, co, 3, 1, d, d, 0, P(sc), P(sh), None, None, P(bits), 69, 0, ctypes.c_float(0.0), 10, st)
        assert rc == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        L.mi_conv_gemm_pp(P(x), P(wp), P(out), B, H, H, ci, H, H, co, 3, 1, d, d, 0, P(sc), P(sh), None, None, P(bits), 69, 0, ctypes.c_float(0.0), 10, st)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print("3x3 %d d2 f69 (stamped build): %.1f us  %.0f TF" % (ci, us, 2.0 * B * H * H * ci * co * 9 / us / 1e6))
    STEPS, PTS = 80, 6
    buf = (ctypes.c_uint * (2 * STEPS * PTS))()
    assert L.mi_pp_trace_read(buf, 2 * STEPS * PTS) == 0
    clk = (ctypes.c_uint * 4)()
    assert L.mi_pp_clock_read(clk) == 0
    print("main loop of the stamped workgroup: %d s_memtime ticks in %.2f us (s_memrealtime, 100 MHz) = %.3f GHz" % (clk[0], clk[1] / 100.0, clk[0] / (clk[1] * 10.0)))
    ns = min(STEPS, 9 * ci // 32)
    names = ["DMA issue", "frag read + lgkm wait", "vm wait(G1) + barrier 1", "MFMA segment", "vm wait(G0) + barrier 2", "(loop back)"]
    for g in range(2):
        t = [[buf[(g * STEPS + s) * PTS + k] for k in range(PTS)] for s in range(ns)]
        lo, hi = 8, ns - 6
        print("group %d: k-step period %.0f cycles (steps %d..%d)" % (g, ((t[hi][0] - t[lo][0]) & 0xffffffff) / (hi - lo), lo, hi))
        for k in range(PTS):
            if k < PTS - 1:
                dl = [((t[s][k + 1] - t[s][k]) & 0xffffffff) for s in range(lo, hi)]
            else:
                dl = [((t[s + 1][0] - t[s][k]) & 0xffffffff) for s in range(lo, hi)]
            print("   %-26s avg %6.0f  min %5d  max %5d" % (names[k], sum(dl) / len(dl), min(dl), max(dl)))
    t0 = [buf[(0 * STEPS + s) * PTS + 0] for s in range(ns)]
    t1 = [buf[(1 * STEPS + s) * PTS + 0] for s in range(ns)]
    print("group 1 top-of-step lags group 0 by %.0f cycles" % (sum(((b - a) & 0xffffffff) for a, b in zip(t0[8:40], t1[8:40])) / 32))


if __name__ == "__main__":
    build(sys.argv[2:]) if sys.argv[1:2] == ["build"] else main()
