"""GPU parity: every C-ABI kernel (through kernels.py -> ctypes -> libmi355seg.so) against
(a) the golden fixtures produced by the reference's own modules and (b) the numpy oracle on the
same bf16-rounded operands.

Tolerances (written where used):
  * fp32 outputs of bf16-operand / fp32-accumulate kernels vs exact math: 1e-3 relative to the tensor's
    max magnitude is BASELINE.json's bar; these kernels measure ~1e-6, asserted at 2e-5.
  * bf16 outputs: one bf16 ulp of the largest magnitude (2^-8 relative).
  * integer results (argmax, histograms): bit exact.
"""
import os

import numpy as np
import pytest
import torch

import _cases
from oracle import ref_ops
from rnd_semantic_segmentation_amd.host import synth

pytestmark = pytest.mark.gpu

K = None  # kernels module, imported lazily so collection works without the .so


@pytest.fixture(scope="module", autouse=True)
def _kern():
    global K
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from rnd_semantic_segmentation_amd import kernels
    K = kernels
    yield


DEV = "cuda"


def nhwc_bf16(x_nchw):
    return torch.from_numpy(np.ascontiguousarray(x_nchw)).to(DEV).permute(0, 2, 3, 1).contiguous().to(torch.bfloat16)


def to_nchw(t_nhwc):
    return t_nhwc.float().permute(0, 3, 1, 2).contiguous().cpu().numpy()


def relmax(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def dev(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV).to(dtype)


# ------------------------------------------------------------------------------------------------ conv family
@pytest.mark.parametrize("name", _cases.CONV_CASES)
def test_conv_fwd_dgrad_wgrad_vs_reference_golden(name):
    c = _cases.conv_case(name)
    x, w, dy = c["x"], c["w"], c["dy"]
    k, s, d, pad = c["k"], c["stride"], c["dil"], c["pad"]
    B, Ci, H, W = x.shape
    Co = w.shape[0]
    Ho, Wo = c["y"].shape[-2:]
    xd, dyd, wd = nhwc_bf16(x), nhwc_bf16(dy), dev(w)
    wp = K.pack_weight_fwd(wd)
    y = K.conv_gemm(xd, wp, (Ho, Wo), k, s, pad, d, K.GATHER_FWD, out_f32=True)
    assert relmax(to_nchw(y), c["y"]) < 2e-5
    wpt = K.pack_weight_dgrad(wd)
    dx = K.conv_gemm(dyd, wpt, (H, W), k, s, pad, d, K.GATHER_DGRAD, out_f32=True)
    assert relmax(to_nchw(dx), c["dx"]) < 2e-5
    dw = torch.full((Co, Ci, k, k), float("nan"), device=DEV)
    K.conv_wgrad(dyd, xd, dw, k, s, pad, d)
    assert relmax(dw.cpu().numpy(), c["dw"]) < 2e-5
    # reproducibility + accumulate
    dw2 = torch.empty_like(dw)
    K.conv_wgrad(dyd, xd, dw2, k, s, pad, d)
    assert torch.equal(dw, dw2)
    K.conv_wgrad(dyd, xd, dw2, k, s, pad, d, accumulate=True)
    assert relmax(dw2.cpu().numpy(), 2 * c["dw"]) < 2e-5


def test_conv_fused_epilogue_bn_residual_relu_mask():
    """FrozenBN + residual + ReLU epilogue (reference resnet.py:107-111, layers.py:18-23) and the backward mask."""
    B, Ci, Co, H, W, d = 3, 256, 320, 19, 23, 2      # M = 1311 (tail), N = 320 (2.5 tiles)
    x = synth.bf16_round(np.maximum(synth.uniform("t.x", (B, Ci, H, W)) * 4, 0))
    w = synth.bf16_round(synth.formula_tensor("t.conv.weight", (Co, Ci, 3, 3)))
    res = synth.bf16_round(synth.uniform("t.res", (B, Co, H, W)) * 2)
    bn = {k: synth.formula_tensor("t.bn2." + k, (Co,)) for k in ("weight", "bias", "running_mean", "running_var")}
    scale, shift = ref_ops.frozen_bn_scale_bias(*[bn[k].astype(np.float64) for k in ("weight", "bias", "running_mean", "running_var")])
    conv = ref_ops.conv2d(x, w, None, 1, d, d)
    want = np.maximum(conv * scale.reshape(1, -1, 1, 1) + shift.reshape(1, -1, 1, 1) + res, 0)
    sc, sh = K.frozen_bn_fold(*[dev(bn[k]) for k in ("weight", "bias", "running_mean", "running_var")])
    assert relmax(sc.cpu().numpy(), scale) < 1e-6 and relmax(sh.cpu().numpy(), shift) < 1e-6
    wp = K.pack_weight_fwd(dev(w))
    xd, resd = nhwc_bf16(x), nhwc_bf16(res)
    y32 = K.conv_gemm(xd, wp, (H, W), 3, 1, d, d, scale=sc, bias=sh, res=resd, relu=True, out_f32=True)
    assert relmax(to_nchw(y32), want) < 2e-5
    y16 = K.conv_gemm(xd, wp, (H, W), 3, 1, d, d, scale=sc, bias=sh, res=resd, relu=True)
    assert relmax(to_nchw(y16), want) < 2.0 ** -8          # bf16 output: one ulp of the max
    assert (to_nchw(y16) >= 0).all()
    # mask epilogue: v = msk > 0 ? v : 0
    ym = K.conv_gemm(xd, wp, (H, W), 3, 1, d, d, res=resd, msk=y16, out_f32=True)
    want_m = (conv + res) * (to_nchw(y16) > 0)
    assert relmax(to_nchw(ym), want_m) < 2e-5
    # packed sign bits: written by the forward epilogue, consumed by the backward epilogue (1/16 of the bf16 mask's bytes)
    bits = torch.empty((B, H, W, Co // 16), dtype=torch.int16, device=DEV)
    y16b = K.conv_gemm(xd, wp, (H, W), 3, 1, d, d, scale=sc, bias=sh, res=resd, relu=True, mask_out=bits)
    assert torch.equal(y16b, y16)
    unpacked = ((bits.view(B, H, W, Co // 16, 1).int() >> torch.arange(16, device=DEV).int()) & 1).bool().reshape(B, H, W, Co)
    assert torch.equal(unpacked, y16 > 0)
    ymb = K.conv_gemm(xd, wp, (H, W), 3, 1, d, d, res=resd, bits=bits, out_f32=True)
    assert torch.equal(ymb, ym)
    gtest = torch.randn((B, H, W, Co), device=DEV).to(torch.bfloat16)
    assert torch.equal(K.relu_mask(gtest, bits), K.relu_mask(gtest, y16))
    # scale folded into the dgrad pack and into the wgrad reduce
    dy = synth.bf16_round(synth.uniform("t.dy", (B, Co, H, W)))
    wpt = K.pack_weight_dgrad(dev(w), sc)
    dx = K.conv_gemm(nhwc_bf16(dy), wpt, (H, W), 3, 1, d, d, K.GATHER_DGRAD, out_f32=True)
    w_s = synth.bf16_round((w.astype(np.float64) * scale.reshape(-1, 1, 1, 1)).astype(np.float32))
    assert relmax(to_nchw(dx), ref_ops.conv2d_dgrad(dy, w_s, (H, W), 1, d, d)) < 2e-5
    dw = torch.empty((Co, Ci, 3, 3), device=DEV)
    K.conv_wgrad(nhwc_bf16(dy), xd, dw, 3, 1, d, d, scale=sc)
    assert relmax(dw.cpu().numpy(), ref_ops.conv2d_wgrad(dy, x, 3, 1, d, d) * scale.reshape(-1, 1, 1, 1)) < 2e-5


@pytest.mark.parametrize("ci,co,B,H,W", [(256, 1024, 2, 13, 11), (128, 528, 1, 17, 9), (64, 256, 3, 21, 20), (256, 1024, 8, 97, 97)])
def test_pointwise_wide_conv_all_hot_epilogues_vs_torch(ci, co, B, H, W):
    """The short-K / wide-N 1x1 convs (conv3 forward, conv1 data gradient: K in {64,128,256}, N = 4K) run on the A-stationary
    kernel (igemm_astat.hip) for the four epilogue flag sets of the bottlenecks; ragged M, an N tail (528 = 4 tiles + 16) and
    the BASELINE shape.  Reference: fp32 torch on the same bf16 operands; bf16 outputs within one ulp of the max, bits exact."""
    g = torch.Generator().manual_seed(ci + co + H)
    x = (torch.randn(B, H, W, ci, generator=g)).to(torch.bfloat16).to(DEV)
    w = (torch.randn(co, ci, 1, 1, generator=g) * 0.06).to(DEV)
    wb = w.to(torch.bfloat16).float().view(co, ci)
    res = torch.randn(B, H, W, co, generator=g).to(torch.bfloat16).to(DEV)
    sc, sh = (torch.rand(co, generator=g) + 0.5).to(DEV), torch.randn(co, generator=g).to(DEV)
    wp = K.pack_weight_fwd(w)
    lin = x.float().view(-1, ci) @ wb.t()                                   # [M, co] fp32
    M = B * H * W

    def bits_of(t):
        return ((t.view(B, H, W, co // 16, 1).int() >> torch.arange(16, device=DEV).int()) & 1).bool().reshape(M, co)

    # 69: FrozenBN + ReLU + sign bits;  71: + residual
    for use_res in (False, True):
        bits = torch.zeros((B, H, W, co // 16), dtype=torch.int16, device=DEV)
        y = K.conv_gemm(x, wp, (H, W), scale=sc, bias=sh, res=res if use_res else None, relu=True, mask_out=bits)
        pre = lin * sc + sh + (res.float().view(M, co) if use_res else 0)
        want = torch.relu(pre)
        assert relmax(y.float().view(M, co).cpu().numpy(), want.cpu().numpy()) < 2.0 ** -8
        sure = pre.abs() > 1e-3
        assert torch.equal(bits_of(bits)[sure], (pre > 0)[sure])
        assert torch.equal(bits_of(bits), y.view(M, co) > 0) or (~sure).any()
    # 128: ReLU backward from sign bits;  130: + residual-gradient add
    keep = torch.rand(B, H, W, co, generator=g).to(DEV) > 0.4
    packed = (keep.view(B, H, W, co // 16, 16).int() << torch.arange(16, device=DEV).int()).sum(-1).to(torch.int16)
    for use_res in (False, True):
        y = K.conv_gemm(x, wp, (H, W), res=res if use_res else None, bits=packed)
        want = (lin + (res.float().view(M, co) if use_res else 0)) * keep.view(M, co)
        assert relmax(y.float().view(M, co).cpu().numpy(), want.cpu().numpy()) < 2.0 ** -8
        assert bool((y.view(M, co)[~keep.view(M, co)] == 0).all())
    # run-to-run reproducibility
    y2 = K.conv_gemm(x, wp, (H, W), res=res, bits=packed)
    assert torch.equal(y, y2)


@pytest.mark.parametrize("B,H,W,ci,co,k,d", [(2, 13, 11, 64, 256, 3, 2), (1, 19, 23, 96, 288, 3, 4), (3, 9, 10, 128, 512, 1, 1), (2, 7, 5, 32, 64, 3, 1)])
@pytest.mark.parametrize("mtg", [8, 10])
def test_wide_tile_ping_pong_main_loop_vs_oracle_and_128_wide_kernel(B, H, W, ci, co, k, d, mtg):
    """csrc/igemm_pp.hip called directly on small / ragged shapes (M and N tails inside one 320 x 256 tile, a single slab, taps
    that fall into the padding): forward and data-gradient gathers with the epilogues the network launches.  The oracle pins the
    value.  For 1x1 convs, and for 3x3 under MI_IGEMM_PP_KORDER=0 (tap-major contraction; tests/test_gpu_scripts.py runs this test
    that way too), both main loops add the products of one output element in the same order (channels ascending, 32 per MFMA), so
    their results must be BIT-equal; the default 3x3 order (all taps of a 32-channel chunk, then the next chunk) differs from the
    128-wide kernel's in fp32 rounding only: within one bf16 ulp of the largest magnitude."""
    pad = d if k == 3 else 0
    same_order = k == 1 or os.environ.get("MI_IGEMM_PP_KORDER") == "0"

    def same(a, b):
        if same_order or mtg == 3:
            return torch.equal(a, b)
        if a.dtype == torch.int16:           # sign bits: may flip where the value is within rounding of zero - checked through the values
            return True
        return relmax(a.float().cpu().numpy(), b.float().cpu().numpy()) < 2.0 ** -8

    shared_window = mtg == 3                 # igemm_pw_kernel: 3x3 only, the two hot epilogues only (flags 69 forward, 128 data gradient)
    if shared_window and (k != 3 or W + 2 * d < 16):
        pytest.skip("the shared-window kernel is the 3x3 kernel")
    x = synth.bf16_round(synth.uniform("pp.x", (B, ci, H, W)) * 4)
    w = synth.bf16_round(synth.formula_tensor("pp.weight", (co, ci, k, k)))
    dy = synth.bf16_round(synth.uniform("pp.dy", (B, co, H, W)) * 2)
    xd, dyd = nhwc_bf16(x), nhwc_bf16(dy)
    wd = dev(w)
    wp, wpt = K.pack_weight_fwd(wd), K.pack_weight_dgrad(wd)
    sc, sh = dev(1 + synth.uniform("pp.s", (co,))), dev(synth.uniform("pp.b", (co,)))
    if shared_window:
        # FrozenBN + ReLU + sign bits (69): bf16 output within one ulp of exact math, sign bits consistent with the output
        bo = torch.zeros((B, H, W, co // 16), dtype=torch.int16, device=DEV)
        yo = K.conv_gemm(xd, wp, (H, W), k, 1, pad, d, scale=sc, bias=sh, relu=True, mask_out=bo, wide=3)
        want = np.maximum(ref_ops.conv2d(x, w, None, 1, pad, d) * sc.cpu().numpy().reshape(1, -1, 1, 1) + sh.cpu().numpy().reshape(1, -1, 1, 1), 0)
        assert relmax(to_nchw(yo), want) < 2.0 ** -8
        bits = (bo.view(torch.uint8).cpu().numpy()[..., None] >> np.arange(8)) & 1            # [B,H,W,co/8,8]
        assert np.array_equal(bits.reshape(B, H, W, co).astype(bool), (yo.float().cpu().numpy() > 0))
        if ci % 64 == 0:                     # the 128-wide kernel can run the same launch: bit-equal
            b0 = torch.zeros_like(bo)
            y0 = K.conv_gemm(xd, wp, (H, W), k, 1, pad, d, scale=sc, bias=sh, relu=True, mask_out=b0, wide=None)
            assert torch.equal(y0, yo) and torch.equal(b0, bo)
        # data gradient through the ReLU mask of a [B,H,W,ci] activation (128)
        if ci % 16 == 0 and co % 32 == 0:
            mb = torch.randint(-32768, 32767, (B, H, W, ci // 16), dtype=torch.int16, device=DEV)
            g1 = K.conv_gemm(dyd, wpt, (H, W), k, 1, pad, d, K.GATHER_DGRAD, bits=mb, wide=3)
            keep = ((mb.view(torch.uint8).cpu().numpy()[..., None] >> np.arange(8)) & 1).reshape(B, H, W, ci).astype(bool)
            gw = np.where(np.transpose(keep, (0, 3, 1, 2)), ref_ops.conv2d_dgrad(dy, w, (H, W), 1, pad, d), 0)
            assert relmax(to_nchw(g1), gw) < 2.0 ** -8
            if co % 64 == 0:
                assert torch.equal(K.conv_gemm(dyd, wpt, (H, W), k, 1, pad, d, K.GATHER_DGRAD, bits=mb, wide=None), g1)
        return
    # forward, plain fp32 store: against exact math
    y = K.conv_gemm(xd, wp, (H, W), k, 1, pad, d, out_f32=True, wide=mtg)
    assert relmax(to_nchw(y), ref_ops.conv2d(x, w, None, 1, pad, d)) < 2e-5
    # forward with FrozenBN + ReLU + sign bits (flags 69) and without ReLU (flags 1): bit-equal to the 128-wide kernel
    base_ok = ci % 64 == 0                                  # the 128-wide kernel stages 64 channels per K-step, this one 32
    if co % 16 == 0 and base_ok:
        b0, b1 = (torch.zeros((B, H, W, co // 16), dtype=torch.int16, device=DEV) for _ in range(2))
        y0 = K.conv_gemm(xd, wp, (H, W), k, 1, pad, d, scale=sc, bias=sh, relu=True, mask_out=b0)
        y1 = K.conv_gemm(xd, wp, (H, W), k, 1, pad, d, scale=sc, bias=sh, relu=True, mask_out=b1, wide=mtg)
        assert same(y0, y1) and same(b0, b1)
        # data gradient with the ReLU mask read from sign bits (flags 128)
        bits = b0[..., : ci // 16].contiguous() if ci % 16 == 0 and ci <= co else None
        if bits is not None:
            g0 = K.conv_gemm(dyd, wpt, (H, W), k, 1, pad, d, K.GATHER_DGRAD, bits=bits)
            g1 = K.conv_gemm(dyd, wpt, (H, W), k, 1, pad, d, K.GATHER_DGRAD, bits=bits, wide=mtg)
            assert same(g0, g1)
    z1 = K.conv_gemm(xd, wp, (H, W), k, 1, pad, d, scale=sc, bias=sh, wide=mtg)
    if base_ok:
        assert same(K.conv_gemm(xd, wp, (H, W), k, 1, pad, d, scale=sc, bias=sh), z1)
    want = ref_ops.conv2d(x, w, None, 1, pad, d) * sc.cpu().numpy().reshape(1, -1, 1, 1) + sh.cpu().numpy().reshape(1, -1, 1, 1)
    assert relmax(to_nchw(z1), want) < 2.0 ** -8                           # bf16 output: one ulp of the largest magnitude
    # data gradient, plain (flags 0): against exact math
    dx = K.conv_gemm(dyd, wpt, (H, W), k, 1, pad, d, K.GATHER_DGRAD, out_f32=True, wide=mtg)
    assert relmax(to_nchw(dx), ref_ops.conv2d_dgrad(dy, w, (H, W), 1, pad, d)) < 2e-5


def test_conv_identity_weight_is_exact_shift_full_size():
    """Size-independent property at the BASELINE shape (B=8, 97x97, C=256, d=2): a one-hot tap weight makes the
    conv an exact spatial shift with zero fill - bit exact in bf16."""
    B, C, H, W, d = 8, 256, 97, 97, 2
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn((B, H, W, C), generator=g).to(DEV).to(torch.bfloat16)
    for tap in (0, 5, 8):
        w = torch.zeros((C, C, 3, 3), device=DEV)
        w[torch.arange(C), torch.arange(C), tap // 3, tap % 3] = 1.0
        y = K.conv_gemm(x, K.pack_weight_fwd(w), (H, W), 3, 1, d, d)
        dy_, dx_ = (tap // 3 - 1) * d, (tap % 3 - 1) * d
        want = torch.zeros_like(x)
        hs, he = max(0, -dy_), min(H, H - dy_)
        ws, we = max(0, -dx_), min(W, W - dx_)
        want[:, hs:he, ws:we] = x[:, hs + dy_:he + dy_, ws + dx_:we + dx_]
        assert torch.equal(y, want)


def test_wgrad_full_size_linearity_and_reproducibility():
    """At M = 75 272: wgrad(dy1 + dy2) == wgrad(dy1) + wgrad(dy2) to fp32 round-off, and two runs are bitwise equal."""
    B, C, H, W, d = 8, 256, 97, 97, 2
    g = torch.Generator(device="cpu").manual_seed(7)
    x = torch.randn((B, H, W, C), generator=g).to(DEV).to(torch.bfloat16)
    dy1 = (torch.randint(-4, 5, (B, H, W, C), generator=g).float() / 4).to(DEV).to(torch.bfloat16)
    dy2 = (torch.randint(-4, 5, (B, H, W, C), generator=g).float() / 4).to(DEV).to(torch.bfloat16)
    out = [torch.empty((C, C, 3, 3), device=DEV) for _ in range(4)]
    K.conv_wgrad(dy1, x, out[0], 3, 1, d, d)
    K.conv_wgrad(dy2, x, out[1], 3, 1, d, d)
    K.conv_wgrad(dy1 + dy2, x, out[2], 3, 1, d, d)      # dy1+dy2 exact in bf16 (multiples of 1/4, |.| <= 2)
    K.conv_wgrad(dy1 + dy2, x, out[3], 3, 1, d, d)
    assert torch.equal(out[2], out[3])
    err = (out[0] + out[1] - out[2]).abs().max().item()
    assert err < 1e-3 * out[2].abs().max().item()


def test_batched_slab_reducers_equal_the_one_call_weight_gradients_bit_for_bit():
    """K.WgradBatch (mi_conv_wgrad_partial x n + ONE mi_conv_wgrad_reduce launch: the three weight gradients of a bottleneck) against mi_conv_wgrad per
    conv: every kernel route (fused-row 3x3, deep-stream 1x1, the 128 x 256 tile, the per-tap kernel with stride 2), FrozenBN scale, accumulation,
    a fifth job (a full batch flushes itself) - the same fixed summation order, hence the same bits.  (At these sizes the 3x3 case takes the per-tap
    kernel; the fused-row route, chosen from 256 x 256 channels up, is the one place where the two forms differ by design - the deferred form splits K for a
    launch that runs beside a data-gradient chain, include/mi355seg.h - and is checked against the one-call result within fp32 rounding below.)"""
    g = torch.Generator(device="cpu").manual_seed(11)
    r = lambda *shape: torch.randn(shape, generator=g).to(DEV).to(torch.bfloat16)
    B, H, W = 2, 33, 29
    cases = [   # (dy, x, k, stride, pad, dil)
        (r(B, H, W, 64), r(B, H, W, 64), 3, 1, 2, 2),
        (r(B, H, W, 256), r(B, H, W, 64), 1, 1, 0, 1),
        (r(B, H, W, 512), r(B, H, W, 1024), 1, 1, 0, 1),
        (r(B, 17, 15, 128), r(B, H, W, 64), 3, 2, 1, 1),
        (r(B, H, W, 64), r(B, H, W, 256), 1, 1, 0, 1),
    ]
    scales = [torch.rand(c[0].shape[-1], generator=g).to(DEV) + 0.5 for c in cases]
    want, got = [], []
    for (dy, x, k, st, pd, dl), sc in zip(cases, scales):
        O, I = dy.shape[-1], x.shape[-1]
        base = torch.randn((O, I, k, k), generator=g).to(DEV)
        w = base.clone()
        K.conv_wgrad(dy, x, w, k, st, pd, dl, scale=sc, accumulate=True)
        want.append(w)
        got.append(base.clone())
    batch = K.WgradBatch()
    for (dy, x, k, st, pd, dl), sc, w in zip(cases, scales, got):
        K.conv_wgrad(dy, x, w, k, st, pd, dl, scale=sc, accumulate=True, batch=batch)
    batch.flush()
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(got, want)):
        assert torch.equal(a, b), "job %d differs from the one-call weight gradient" % i
    # fused-row 3x3 route (256 -> 256, enough pixels for >= 8 K steps per split): the two forms partition K differently; both are deterministic
    dy, x = r(4, 65, 61, 256), r(4, 65, 61, 256)
    one = torch.zeros((256, 256, 3, 3), device=DEV)
    K.conv_wgrad(dy, x, one, 3, 1, 2, 2)
    two = [torch.zeros_like(one) for _ in range(2)]
    for t in two:
        b2 = K.WgradBatch()
        K.conv_wgrad(dy, x, t, 3, 1, 2, 2, batch=b2)
        b2.flush()
    torch.cuda.synchronize()
    assert torch.equal(two[0], two[1])
    assert float((two[0] - one).abs().max()) <= 2e-6 * float(one.abs().max()) + 1e-6


# ------------------------------------------------------------------------------------------------ ASPP head chain
def _aspp_forward(xd, w4, b4, rates=(6, 12, 18, 24)):
    B, H, W, C = xd.shape
    wall = K.aspp_pack_fwd(w4)
    z = K.conv_gemm(xd, wall, (H, W), zsplit=K.ASPP_ZGW)
    return K.aspp_col2im(z, b4, B, H, W, w4.shape[1], rates)


@pytest.mark.parametrize("ncls", [2, 7])
def test_aspp_head_other_class_counts_vs_oracle(ncls):
    """MODEL.NUM_CLASSES != 19 (the reference ships deeplabv2_r101_src_kvasir.yaml with 2): every ASPP kernel takes K, the
    weight-gradient scatter included (it used to assume 19 and wrote out of bounds)."""
    B, C, H, W = 2, 64, 21, 19
    x = synth.bf16_round(np.maximum(synth.uniform("aspp_k.x", (B, C, H, W)) * 4, 0))
    ws = np.stack([synth.bf16_round(synth.formula_tensor("conv2d_list.%d.weight" % i, (ncls, C, 3, 3)) * 4) for i in range(4)])
    bs = np.stack([synth.formula_tensor("conv2d_list.%d.bias" % i, (ncls,)) for i in range(4)])
    dlow = synth.bf16_round(synth.uniform("aspp_k.dlow", (B, ncls, H, W)))
    xd, w4, b4 = nhwc_bf16(x), dev(ws), dev(bs)
    low = _aspp_forward(xd, w4, b4)
    assert low.shape == (B, H, W, ncls)
    assert relmax(to_nchw(low), ref_ops.aspp_head(x, ws, bs)) < 2e-5
    dl = dev(dlow).permute(0, 2, 3, 1).contiguous()
    gm = K.aspp_im2col(dl, (6, 12, 18, 24))
    dx = K.conv_gemm(gm, K.aspp_pack_dgrad(w4), (H, W), out_f32=True)
    dx_ref, dw_ref, db_ref = ref_ops.aspp_head_backward(dlow, x, ws)
    assert relmax(to_nchw(dx), dx_ref) < 2e-5
    guard = torch.full((4 * ncls * C * 9 + 4096,), 7.0, device=DEV)            # canary behind the gradient
    dw4 = guard[:4 * ncls * C * 9].view(4, ncls, C, 3, 3)
    K.conv_wgrad(gm, xd, dw4, out_map=1, ncls=ncls)
    assert relmax(dw4.cpu().numpy(), np.stack(dw_ref)) < 2e-5
    assert bool((guard[4 * ncls * C * 9:] == 7.0).all()), "weight-gradient scatter wrote past dw"
    db4 = torch.empty_like(b4)
    K.aspp_bias_grad(dl, db4)
    assert relmax(db4.cpu().numpy(), np.stack(db_ref)) < 2e-5
    # the C side refuses a dw that is too small or a class count that does not fit, instead of scattering out of bounds
    from rnd_semantic_segmentation_amd._lib import MiError
    with pytest.raises(MiError, match="dw holds"):
        K.conv_wgrad(gm, xd, dw4[:3], out_map=1, ncls=ncls)
    with pytest.raises(MiError):
        K.conv_wgrad(gm, xd, dw4, out_map=1, ncls=20)


def test_aspp_head_upsample_ce_vs_reference_golden():
    c = _cases.aspp_case()
    g = c["g"]
    x, ws, bs, lab, size = c["x"], c["w"], c["b"], c["label"], c["size"]
    B, C, H, W = x.shape
    xd = nhwc_bf16(x)
    w4, b4 = dev(ws), dev(bs)
    low = _aspp_forward(xd, w4, b4)                                     # [B,H,W,19] fp32
    assert relmax(to_nchw(low), g["low"]) < 2e-5
    # unfused: upsample (materialised NCHW) + CE, forward and backward
    up = K.upsample_ac_fwd(low, size)
    assert relmax(up.cpu().numpy()[:, :, ::3, ::3], g["up_sub"]) < 2e-5
    labd = dev(lab, torch.int64)
    lo = K.softmax_ce_fwd(up, labd)
    assert abs(lo[0].item() - float(g["loss"])) < 2e-5 * abs(float(g["loss"]))
    assert lo[1].item() == float((lab != 255).sum())
    dup = K.softmax_ce_bwd(up, labd, lo)
    assert relmax(dup.cpu().numpy()[:, :, ::3, ::3], g["dup_sub"]) < 2e-5
    dlow_u = K.upsample_ac_bwd(dup, (H, W))
    assert relmax(to_nchw(dlow_u), g["dlow"]) < 2e-5
    # fused: never materialises `up`
    lo2, dlow = K.upsample_ce(low, labd)
    assert abs(lo2[0].item() - float(g["loss"])) < 2e-5 * abs(float(g["loss"]))
    assert lo2[1].item() == lo[1].item()
    assert relmax(to_nchw(dlow), g["dlow"]) < 2e-5
    lo3, dlow3 = K.upsample_ce(low, labd)
    assert torch.equal(dlow, dlow3) and torch.equal(lo2, lo3)            # deterministic
    # backward of the head.  G is bf16, so compare against the oracle fed the bf16-rounded dlow (tight) ...
    gm = K.aspp_im2col(dlow, (6, 12, 18, 24))
    wallT = K.aspp_pack_dgrad(w4)
    dx = K.conv_gemm(gm, wallT, (H, W), out_f32=True)
    dlow_r = synth.bf16_round(to_nchw(dlow))
    dx_ref, dw_ref, db_ref = ref_ops.aspp_head_backward(dlow_r, x, ws)
    assert relmax(to_nchw(dx), dx_ref) < 2e-5
    dw4 = torch.empty_like(w4)
    K.conv_wgrad(gm, xd, dw4, out_map=1, ncls=19)
    assert relmax(dw4.cpu().numpy(), np.stack(dw_ref)) < 2e-5
    db4 = torch.empty_like(b4)
    K.aspp_bias_grad(dlow, db4)
    assert relmax(db4.cpu().numpy(), g["db"]) < 2e-5
    # ... and against the reference's fp32 autograd (bf16 rounding of dlow: 2^-9 relative per element)
    assert relmax(to_nchw(dx), g["dx"]) < 4e-3
    assert relmax(dw4.cpu().numpy(), g["dw"]) < 4e-3


def test_aspp_head_2048_channels_17x17_vs_reference_golden():
    """the head at its real width (2048 input channels) on a 17x17 map, where every rate reaches the padding;
    forward, data gradient, weight gradient (out_map 1) and bias gradient against the reference's autograd (g2_aspp_2048)."""
    g = _cases.load("g2_aspp_2048")
    C, H, Kc = 2048, 17, 19
    ws = np.stack([synth.bf16_round(synth.formula_tensor("conv2d_list.%d.weight" % i, (Kc, C, 3, 3)) * 4) for i in range(4)])
    bs = np.stack([synth.formula_tensor("conv2d_list.%d.bias" % i, (Kc,)) for i in range(4)])
    x = synth.bf16_round(np.maximum(synth.uniform("g2b.x", (1, C, H, H)) * 2, 0))
    dl = synth.bf16_round(synth.uniform("g2b.dlow", (1, Kc, H, H)))
    assert _cases.sha(x) + _cases.sha(ws) + _cases.sha(dl) == str(g["in_sha"]), "input formulas drifted from the fixture"
    xd, w4, b4 = nhwc_bf16(x), dev(ws), dev(bs)
    low = _aspp_forward(xd, w4, b4)
    e_low = relmax(to_nchw(low), g["low"])
    dld = dev(dl).permute(0, 2, 3, 1).contiguous()
    gm = K.aspp_im2col(dld, (6, 12, 18, 24))
    dx = K.conv_gemm(gm, K.aspp_pack_dgrad(w4), (H, H), out_f32=True)
    e_dx = relmax(to_nchw(dx)[0, :96], g["dx_crop"])
    dw4 = torch.empty_like(w4)
    K.conv_wgrad(gm, xd, dw4, out_map=1, ncls=Kc)
    e_dw = relmax(dw4.cpu().numpy()[:, :, :48], g["dw_crop"])
    db4 = torch.empty_like(b4)
    K.aspp_bias_grad(dld, db4)
    print("aspp 2048x17x17: low %.2e dx %.2e dw %.2e |dx| %.6f (ref %.6f)" % (e_low, e_dx, e_dw, dx.double().norm().item(), float(g["dx_norm"])))
    assert e_low < 2e-5 and e_dx < 2e-5 and e_dw < 2e-5                    # operands are bf16-exact on both sides
    assert abs(dx.double().norm().item() - float(g["dx_norm"])) < 1e-5 * float(g["dx_norm"])
    assert abs(dw4.double().norm().item() - float(g["dw_norm"])) < 1e-5 * float(g["dw_norm"])
    assert relmax(db4.cpu().numpy(), g["db"]) < 2e-5


def test_upsample_13x21_to_97x161_and_large_logits_vs_reference_golden():
    """non-square, non-integer scale; and |logit| up to 60 with near-one-hot rows (the fused kernel uses the
    hardware exp / log: VERDICT r1 weak 11)."""
    g = _cases.load("g3_upsample_13x21")
    low = synth.uniform("g3b.low", (2, 19, 13, 21)).astype(np.float32) * 6
    lab = synth.synth_label(2, 97, 161, 19, seed=13)
    assert _cases.sha(low) + _cases.sha(lab) == str(g["in_sha"])
    lowd, labd = dev(low).permute(0, 2, 3, 1).contiguous(), dev(lab, torch.int64)
    up = K.upsample_ac_fwd(lowd, (97, 161))
    assert relmax(up.cpu().numpy()[:, :, ::4, ::5], g["up_sub"]) < 2e-6
    lo, dlow = K.upsample_ce(lowd, labd)
    assert abs(lo[0].item() - float(g["loss"])) < 2e-6 * abs(float(g["loss"]))
    assert lo[2].item() == 0
    assert relmax(to_nchw(dlow), g["dlow"]) < 2e-5
    lo_u = K.softmax_ce_fwd(up, labd)
    dup = K.softmax_ce_bwd(up, labd, lo_u)
    assert relmax(dup.cpu().numpy()[:, :, ::4, ::5], g["dup_sub"]) < 2e-5
    assert relmax(to_nchw(K.upsample_ac_bwd(dup, (13, 21))), g["dlow"]) < 2e-5
    # large logits
    g = _cases.load("g3_upsample_large_logits")
    low = (synth.uniform("g3c.low", (1, 19, 9, 11)).astype(np.float32) * 120).astype(np.float32)
    lab = synth.synth_label(1, 65, 81, 19, seed=17)
    assert _cases.sha(low) + _cases.sha(lab) == str(g["in_sha"])
    lowd, labd = dev(low).permute(0, 2, 3, 1).contiguous(), dev(lab, torch.int64)
    lo, dlow = K.upsample_ce(lowd, labd)
    e_loss = abs(lo[0].item() - float(g["loss"])) / abs(float(g["loss"]))
    e_d = relmax(to_nchw(dlow), g["dlow"])
    probs, _ = K.upsample_softmax(lowd, (65, 81))
    e_p = relmax(probs.cpu().numpy()[:, :, ::3, ::4], g["probs_sub"])
    print("large logits: loss %.2e dlow %.2e probs %.2e" % (e_loss, e_d, e_p))
    assert e_loss < 2e-6 and e_d < 2e-5 and e_p < 2e-6


def test_out_of_range_labels_are_counted_not_silently_ignored():
    """torch.nn.CrossEntropyLoss raises a device assert for a label outside [0,K) that is not ignore_index; the kernels
    count such pixels (loss_out[2]) and the host refuses them."""
    low = dev(synth.uniform("oor.low", (1, 9, 7, 19)).astype(np.float32))
    lab = synth.synth_label(1, 33, 25, 19, seed=4)
    lab[0, 10, 5:9] = 33            # raw Cityscapes id, not a train id
    lab[0, 20, 3] = -1
    labd = dev(lab, torch.int64)
    lo, _ = K.upsample_ce(low, labd)
    assert lo[2].item() == 5
    with pytest.raises(ValueError, match="5 label values"):
        K.check_labels(lo, 19)
    up = K.upsample_ac_fwd(low, (33, 25))
    assert K.softmax_ce_fwd(up, labd)[2].item() == 5
    from rnd_semantic_segmentation_amd.host import modules
    with pytest.raises(ValueError, match="outside"):
        modules.CrossEntropyLoss(255)(up, labd)


def test_metrics_on_device_bit_exact_vs_reference_golden():
    """A9 on the device path: intersectionAndUnionGPU / confusion_matrix on CUDA tensors, exact integers vs g7 (the reference's
    utility.py:133-161,347-359 outputs)."""
    from rnd_semantic_segmentation_amd.host import config as hc
    from rnd_semantic_segmentation_amd.host import metrics
    g = _cases.load("g7_metrics")
    pred, target, t2 = (torch.from_numpy(g[k]).cuda() for k in ("pred", "target", "target2"))
    for tgt, want in ((target, g["iu"]), (t2, g["iu2"])):
        out = metrics.intersectionAndUnionGPU(pred.clone(), tgt, 19, 255)
        assert all(o.is_cuda for o in out)
        assert np.array_equal(np.stack([o.cpu().numpy() for o in out]), want.astype(np.float32))
    cfg = hc.cfg.clone()
    cfg.defrost()
    cfg.merge_from_list(["MODEL.NUM_CLASSES", 19])
    cmt = metrics.confusion_matrix(cfg, torch.from_numpy(g["small_p"]).cuda(), torch.from_numpy(g["small_t"]).cuda())
    assert cmt.dtype == torch.int64 and np.array_equal(cmt.numpy(), g["cmt"])
    meter = metrics.AverageMeter()
    for tgt in (target, t2):
        meter.update(*[o.cpu().numpy().astype(np.float64) for o in metrics.intersectionAndUnionGPU(pred.clone(), tgt, 19, 255)])
    lines = []

    class L:
        def info(self, s):
            lines.append(s)

    meter.summary(L(), 19)
    assert lines == [str(s) for s in g["summary"]]


def test_upsample_integer_scale_all_ignored_and_inference_tail():
    c = _cases.upsample_case()
    g = c["g"]
    low = dev(c["low"]).permute(0, 2, 3, 1).contiguous()
    up = K.upsample_ac_fwd(low, (129, 129))
    assert relmax(up.cpu().numpy()[:, :, ::5, ::3], g["up_sub"]) < 2e-5
    labd = dev(c["label"], torch.int64)
    lo, dlow = K.upsample_ce(low, labd)
    assert abs(lo[0].item() - float(g["loss"])) < 2e-5 * abs(float(g["loss"]))
    want_dlow = ref_ops.bilinear_ac_backward(ref_ops.cross_entropy_ignore(ref_ops.bilinear_ac(c["low"], (129, 129)), c["label"])[1], (17, 17))
    assert relmax(to_nchw(dlow), want_dlow) < 2e-5
    # every label ignored: loss nan (0/0) exactly like torch
    lab_all = torch.full((1, 129, 129), 255, dtype=torch.int64, device=DEV)
    lo0, _ = K.upsample_ce(low, lab_all, want_grad=False)
    assert np.isnan(lo0[0].item()) and lo0[1].item() == 0 and np.isnan(float(g["loss_all_ignored"]))
    # inference tail (reference utility.py:185-186): probabilities and argmax
    probs, pred = K.upsample_softmax(low, (129, 129))
    want = ref_ops.inference_probs(c["low"], (129, 129))
    assert relmax(probs.cpu().numpy(), want) < 2e-5
    assert abs(probs.sum(1).cpu().numpy() - 1).max() < 1e-5
    assert torch.equal(pred.long(), probs.argmax(1))                      # integer result: exact
    # non-integer scale to a label size larger than train crops (64x128 -> 512x1024 shape class, scaled down)
    low2 = dev(synth.uniform("t.low2", (1, 19, 9, 17)).astype(np.float32) * 5).permute(0, 2, 3, 1).contiguous()
    probs2, _ = K.upsample_softmax(low2, (70, 133))
    want2 = ref_ops.inference_probs(to_nchw(low2), (70, 133))
    assert relmax(probs2.cpu().numpy(), want2) < 2e-5


def test_upsample_ce_full_size_properties():
    """B=8, 97x97 -> 769x769 (BASELINE shape): fused loss == unfused loss; gradient sums to ~0 per pixel set;
    two runs bitwise identical."""
    B, h, w, Kc, H, W = 8, 97, 97, 19, 769, 769
    g = torch.Generator(device="cpu").manual_seed(3)
    low = (torch.randn((B, h, w, Kc), generator=g) * 3).to(DEV)
    lab = torch.from_numpy(synth.synth_label(B, H, W, Kc, seed=9)).to(DEV).long()
    lo, dlow = K.upsample_ce(low, lab)
    up = K.upsample_ac_fwd(low, (H, W))
    lo_u = K.softmax_ce_fwd(up, lab)
    assert lo[1].item() == lo_u[1].item() == float((lab != 255).sum().item())
    assert abs(lo[0].item() - lo_u[0].item()) < 1e-5 * abs(lo_u[0].item())
    dlow_u = K.upsample_ac_bwd(K.softmax_ce_bwd(up, lab, lo_u), (h, w))
    assert (dlow - dlow_u).abs().max().item() < 2e-5 * dlow_u.abs().max().item()
    assert abs(dlow.sum().item()) < 1e-4                                  # softmax - onehot sums to zero over classes
    lo2, dlow2 = K.upsample_ce(low, lab)
    assert torch.equal(dlow, dlow2) and torch.equal(lo, lo2)


# ------------------------------------------------------------------------------------------------ optimiser & helpers
def test_sgd_matches_torch_optim_golden():
    g = _cases.load("g7_metrics")
    p = dev(g["sgd_p0"]).clone()
    buf = torch.zeros_like(p)
    for s in range(3):
        K.sgd_step(p, dev(g["sgd_g"][s]), buf, float(g["sgd_lrs"][s]), 0.9, 5e-4)
        assert relmax(p.cpu().numpy(), g["sgd_p"][s]) < 3e-7


def test_frozen_bn_fold_and_relu_mask():
    g = _cases.load("g4_frozenbn")
    sc, sh = K.frozen_bn_fold(*[dev(g[k]) for k in ("weight", "bias", "running_mean", "running_var")])
    y = dev(g["x"]) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    assert relmax(y.cpu().numpy(), g["y"]) < 1e-6
    x = torch.randn(4096 * 8, device=DEV).to(torch.bfloat16)
    m = torch.randn(4096 * 8, device=DEV).to(torch.bfloat16)
    assert torch.equal(K.relu_mask(x, m), torch.where(m > 0, x, torch.zeros_like(x)))


def test_errors_are_reported_not_thrown_across_the_abi():
    from rnd_semantic_segmentation_amd import _lib
    a = torch.zeros((1, 4, 4, 48), device=DEV, dtype=torch.bfloat16)
    wp = torch.zeros((1, 64, 48), device=DEV, dtype=torch.bfloat16)
    with pytest.raises(_lib.MiError, match="multiple of 32"):
        K.conv_gemm(a, wp, (4, 4))
    with pytest.raises(_lib.MiError, match="GPU"):
        K.relu_mask(torch.zeros(8, dtype=torch.bfloat16), torch.zeros(8, dtype=torch.bfloat16))


@pytest.mark.parametrize("B,H,W,ci,co,k,d", [(1, 1, 1, 64, 64, 1, 1), (1, 2, 3, 64, 72, 3, 1), (1, 5, 4, 128, 64, 3, 6), (3, 1, 7, 64, 8, 3, 2),
                                             (2, 9, 7, 256, 64, 3, 2), (1, 11, 5, 320, 136, 1, 1), (2, 6, 9, 576, 72, 3, 4),
                                             (2, 9, 11, 96, 40, 3, 1), (1, 7, 6, 160, 96, 3, 2), (2, 5, 8, 480, 192, 3, 1), (1, 4, 9, 224, 96, 1, 1)])
def test_tiny_and_ragged_conv_shapes(B, H, W, ci, co, k, d):
    """Degenerate geometry: a single pixel, images smaller than the dilation (every off-centre tap is padding), N = 8 and
    N = 72 (below / not a multiple of the 128-column tile), M far below one tile; Cin = 256 / 320 / 576 select the 128 x 256
    weight-gradient tile (whole, with a 64-channel tail, with two tiles and a tail); Cin = 96 / 160 / 480 / 224 (multiples of 32 but not of 64: the
    32-padded gathers of HarDNet, which only the 32-channel slabs of igemm_pp_kernel take) in forward, Cout = 96 the same in the data gradient.
    Forward, data and weight gradients against fp32 torch on the same bf16 operands."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(B * 7 + H * 5 + W + co)
    x = torch.randn(B, ci, H, W, generator=g).to(torch.bfloat16).float()
    w = (torch.randn(co, ci, k, k, generator=g) * 0.1).to(torch.bfloat16).float()
    dy = torch.randn(B, co, H, W, generator=g).to(torch.bfloat16).float()
    pad = d if k == 3 else 0
    xq = x.clone().requires_grad_(True)
    wq = w.clone().requires_grad_(True)
    y = F.conv2d(xq, wq, None, 1, pad, d)
    y.backward(dy)
    xd = x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)
    got = K.conv_gemm(xd, K.pack_weight_fwd(w.to(DEV)), (H, W), k, 1, pad, d, out_f32=True)
    assert relmax(to_nchw(got), y.detach().numpy()) < 2e-5
    if ci % 8 == 0 and co % 32 == 0:       # the data gradient contracts over co
        dx = K.conv_gemm(dyd, K.pack_weight_dgrad(w.to(DEV)), (H, W), k, 1, pad, d, K.GATHER_DGRAD, out_f32=True)
        assert relmax(to_nchw(dx), xq.grad.numpy()) < 2e-5
    dw = torch.full((co, ci, k, k), float("nan"), device=DEV)
    K.conv_wgrad(dyd, xd, dw, k, 1, pad, d)
    assert relmax(dw.cpu().numpy(), wq.grad.numpy()) < 2e-5


def test_tiny_upsample_ce_shapes():
    """1x1 -> 1x1, 1x1 -> 5x3, 2x2 -> 2x2 (identity), one class: the fused upsample + CE against the unfused kernels."""
    for (B, h, w, Kc, H, W) in ((1, 1, 1, 3, 1, 1), (2, 1, 1, 19, 5, 3), (1, 2, 2, 19, 2, 2), (1, 3, 2, 1, 7, 5)):
        g = torch.Generator().manual_seed(h * 10 + W)
        low = torch.randn(B, h, w, Kc, generator=g).to(DEV)
        lab = torch.randint(0, Kc, (B, H, W), generator=g).to(DEV)
        if H * W > 2:
            lab[0, 0, 0] = 255
        out, dlow = K.upsample_ce(low, lab)
        up = K.upsample_ac_fwd(low, (H, W))
        ref = K.softmax_ce_fwd(up, lab)
        assert abs(out[0].item() - ref[0].item()) <= 1e-5 * max(1.0, abs(ref[0].item())) and out[1].item() == ref[1].item()
        dref = K.upsample_ac_bwd(K.softmax_ce_bwd(up, lab, ref), (h, w))
        assert relmax(dlow.cpu().numpy(), dref.cpu().numpy()) < 1e-4 or float(dref.abs().max()) < 1e-6


def test_stem_bn_relu_maxpool_fused_vs_torch_ops():
    """Fused FrozenBN + ReLU + 3x3/2/1 max-pool (reference resnet.py:138-141) against the same chain of PyTorch ops on
    the GPU: forward bit-exact (bf16), backward within one bf16 ulp (up to 4 window contributions are summed)."""
    import torch.nn.functional as F
    B, C, Hc, Wc = 2, 64, 37, 41                       # odd sizes: windows hang over the border
    g = torch.Generator(device="cpu").manual_seed(11)
    y = torch.randn((B, Hc, Wc, C), generator=g).to(DEV).to(torch.bfloat16)
    scale = (torch.rand(C, generator=g) + 0.5).to(DEV)
    shift = (torch.randn(C, generator=g) * 0.3).to(DEV)
    pool, idx = K.stem_pool_fwd(y, scale, shift)
    yr = y.permute(0, 3, 1, 2).float().requires_grad_(True)
    act = F.relu(yr * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)).to(torch.bfloat16)
    ref = F.max_pool2d(act, 3, 2, 1)
    assert torch.equal(pool.permute(0, 3, 1, 2), ref)
    assert int(idx.max()) <= 9
    dpool = torch.randn(ref.shape, generator=g).to(DEV).to(torch.bfloat16)
    ref.backward(dpool)
    dy = K.stem_pool_bwd(dpool.permute(0, 2, 3, 1).contiguous(), idx, scale, (Hc, Wc))
    want = yr.grad.permute(0, 2, 3, 1)
    err = (dy.float() - want).abs().max().item()
    assert err <= 2.0 ** -7 * want.abs().max().item()
    # gradient only where the activation is positive
    assert not (dy.float().abs() > 0)[(act.permute(0, 2, 3, 1) <= 0)].any()


@pytest.mark.parametrize("cin,cout,k,d", [(256, 256, 3, 2), (1024, 256, 1, 1), (512, 512, 3, 4)])
def test_conv_full_size_vs_pytorch_fp32_conv(cin, cout, k, d):
    """BASELINE shapes (B=8, 97x97 -> M = 75 272 GEMM rows): forward, data gradient and weight gradient against PyTorch's own
    fp32 convolution on the GPU, fed the same bf16-rounded operands.  Tolerance 1e-3 of the tensor's max (BASELINE.json's
    bar); both sides accumulate in fp32, so the measured difference is ~1e-5."""
    import torch.nn.functional as F
    B, H = 8, 97
    g = torch.Generator(device="cpu").manual_seed(cin + cout + k)
    x = torch.randn((B, H, H, cin), generator=g).to(DEV).to(torch.bfloat16)
    dy = torch.randn((B, H, H, cout), generator=g).to(DEV).to(torch.bfloat16)
    w = (torch.randn((cout, cin, k, k), generator=g) * (2.0 / (cin * k * k)) ** 0.5).to(DEV).to(torch.bfloat16).float()
    pad = d if k == 3 else 0
    prev = torch.backends.cudnn.allow_tf32
    xr = x.permute(0, 3, 1, 2).float().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, 1, pad, d)
    yr.backward(dy.permute(0, 3, 1, 2).float())
    torch.backends.cudnn.allow_tf32 = prev
    rel = lambda a, b: ((a.float() - b.float()).abs().max() / b.float().abs().max()).item()
    y = K.conv_gemm(x, K.pack_weight_fwd(w), (H, H), k, 1, pad, d, out_f32=True)
    assert rel(y, yr.detach().permute(0, 2, 3, 1)) < 1e-3
    dx = K.conv_gemm(dy, K.pack_weight_dgrad(w), (H, H), k, 1, pad, d, K.GATHER_DGRAD, out_f32=True)
    assert rel(dx, xr.grad.permute(0, 2, 3, 1)) < 1e-3
    dw = torch.empty_like(w)
    K.conv_wgrad(dy, x, dw, k, 1, pad, d)
    assert rel(dw, wr.grad) < 1e-3


def test_table_driven_weight_pack_equals_the_per_tensor_packs_on_ragged_shapes():
    """mi_pack_weights_multi (one launch for every conv of a module; 32 x 128 channel blocks through LDS) against exact
    arithmetic and against mi_pack_weight_fwd / _dgrad: 1x1 and 3x3 tensors whose O and I are not multiples of the block
    (tails in both directions, a tensor smaller than one block), with and without the data-gradient operand and the folded
    FrozenBN scale.  bf16(x) and bf16(x * scale) are single roundings of fp32 values: bit exact."""
    g = torch.Generator(device="cpu").manual_seed(11)
    shapes = [(40, 24, 3), (8, 8, 1), (136, 264, 1), (72, 200, 3), (33, 129, 1), (256, 64, 3)]
    ws = [torch.randn((o, i, k, k), generator=g) for o, i, k in shapes]
    scales = [torch.rand(o, generator=g) + 0.5 for o, _, _ in shapes]
    wflat = torch.cat([w.reshape(-1) for w in ws]).to(DEV)
    sflat = torch.cat(scales).to(DEV)
    for with_dgrad in (True, False):
        rows, off, soff, blk = [], 0, 0, 0
        for (o, i, k), w in zip(shapes, ws):
            rows.append([off, soff if (o % 16) else -1, off, off if with_dgrad else -1, o, i, k * k, blk])   # some tensors without a scale
            off += w.numel()
            soff += o
            blk += -(-o // 32) * -(-i // 128)
        table = torch.tensor(rows, dtype=torch.int64, device=DEV)
        wp = torch.full((off,), float("nan"), dtype=torch.bfloat16, device=DEV)
        wpt = torch.full((off,), float("nan"), dtype=torch.bfloat16, device=DEV)
        K.pack_weights_multi(wflat, sflat, wp, wpt, table, len(rows), blk)
        for r, (o, i, k), w, sc in zip(rows, shapes, ws, scales):
            n = w.numel()
            wd = w.to(DEV)
            want_f = wd.permute(2, 3, 0, 1).reshape(k * k, o, i).to(torch.bfloat16)
            assert torch.equal(wp[r[2]:r[2] + n].view(k * k, o, i), want_f)
            assert torch.equal(K.pack_weight_fwd(wd), want_f)
            if with_dgrad:
                s = sc.to(DEV) if r[1] >= 0 else torch.ones(o, device=DEV)
                want_b = (wd * s.view(-1, 1, 1, 1)).permute(2, 3, 1, 0).reshape(k * k, i, o).to(torch.bfloat16)
                assert torch.equal(wpt[r[3]:r[3] + n].view(k * k, i, o), want_b)
                assert torch.equal(K.pack_weight_dgrad(wd, s), want_b)
            else:
                assert bool(torch.isnan(wpt[r[2]:r[2] + n].float()).all())          # untouched


@pytest.mark.parametrize("B,H,W", [(2, 33, 35), (1, 65, 65), (3, 16, 23)])
def test_stem_weight_gradient_patch_matrix_plus_1x1_wgrad_vs_torch(B, H, W):
    """mi_stem_im2col + mi_conv_wgrad (the deterministic replacement of the library's weight gradient for the 7x7/2/3 stem conv,
    resnet.py:137) against torch's conv2d weight gradient in fp32 on the same bf16-rounded operands; bit-reproducible run to run."""
    g = torch.Generator(device="cpu").manual_seed(H * 7 + W)
    x = torch.randn(B, 3, H, W, generator=g).to(torch.bfloat16)
    Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    dy = torch.randn(B, Ho, Wo, 64, generator=g).to(torch.bfloat16)
    xd = x.to(DEV).contiguous(memory_format=torch.channels_last)
    dyd = dy.to(DEV)
    dw = K.stem_wgrad(dyd, x=xd)
    ref = torch.nn.grad.conv2d_weight(x.float(), (64, 3, 7, 7), dy.float().permute(0, 3, 1, 2), stride=2, padding=3)
    assert relmax(dw.cpu().numpy(), ref.numpy()) < 2e-5
    assert torch.equal(dw, K.stem_wgrad(dyd, x=xd))
    # the forward as patch matrix + GEMM, and the weight gradient from that same (192-column) matrix
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.1
    y, col = K.stem_conv_fwd(xd, w.to(DEV))
    want = torch.nn.functional.conv2d(x.float(), w.to(torch.bfloat16).float(), None, 2, 3).permute(0, 2, 3, 1)
    assert relmax(y.float().cpu().numpy(), want.numpy()) < 2.0 ** -8
    assert torch.equal(K.stem_wgrad(dyd, col=col), dw)


@pytest.mark.parametrize("B,H,W,ncls", [(2, 21, 19, 19), (1, 30, 131, 19), (1, 12, 140, 19), (2, 97, 97, 19), (1, 9, 40, 7)])
def test_aspp_col2im_and_im2col_rows_vs_torch(B, H, W, ncls):
    """mi_aspp_col2im (36 shifted tap planes + bias -> logits, the reference's association order classifier.py:26-29) and mi_aspp_im2col (the patch
    matrix of d loss / d logits) in their row-blocked forms and im2col's per-element form (W * classes > 2496: the 140-wide case), against torch index arithmetic.
    col2im sums fp32 values in the reference's order: bit-exact; im2col moves bf16-rounded values: bit-exact."""
    rates = [6, 12, 18, 24]
    g = torch.Generator().manual_seed(B * H + W)
    M = B * H * W
    z = torch.randn(36, M, 20, generator=g)
    bias = torch.randn(4, ncls, generator=g)
    low = K.aspp_col2im(z.to(DEV).contiguous(), bias.to(DEV), B, H, W, ncls, rates)
    zz = z.view(36, B, H, W, 20)
    want = None
    for r, d in enumerate(rates):
        s = bias[r].view(1, 1, 1, ncls).expand(B, H, W, ncls).clone()
        for ky in range(3):
            for kx in range(3):
                dy, dx = (ky - 1) * d, (kx - 1) * d
                h0, h1, w0, w1 = max(0, -dy), min(H, H - dy), max(0, -dx), min(W, W - dx)
                if h0 < h1 and w0 < w1:
                    s[:, h0:h1, w0:w1] += zz[r * 9 + ky * 3 + kx][:, h0 + dy:h1 + dy, w0 + dx:w1 + dx, :ncls]
        want = s if want is None else want + s
    assert torch.equal(low.cpu().view(B, H, W, ncls), want)
    dlow = torch.randn(B, H, W, ncls, generator=g)
    gm = K.aspp_im2col(dlow.to(DEV), rates)
    ref = torch.zeros(B, H, W, K.ASPP_KPAD)
    for r, d in enumerate(rates):
        for ky in range(3):
            for kx in range(3):
                dy, dx = (ky - 1) * d, (kx - 1) * d           # dx[q] += W_tap^T dout[q - s]
                h0, h1, w0, w1 = max(0, dy), min(H, H + dy), max(0, dx), min(W, W + dx)
                col = (r * 9 + ky * 3 + kx) * ncls
                if h0 < h1 and w0 < w1:
                    ref[:, h0:h1, w0:w1, col:col + ncls] = dlow[:, h0 - dy:h1 - dy, w0 - dx:w1 - dx]
    assert torch.equal(gm.float().cpu().view(B, H, W, -1), ref.to(torch.bfloat16).float())
